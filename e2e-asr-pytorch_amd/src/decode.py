"""Beam-search decoder with the reference's behaviour (src/decode.py:65-281) — hypothesis expansion, the <eos>
threshold rule, joint CTC / RNN-LM scoring, average-score pruning — batched over utterances AND live hypotheses on the
GPU.  `forward` keeps the whole search on the device: per output position one attention+decoder step, one CTC
prefix-score launch, one LM step and ONE bookkeeping kernel (asr_beam_step: fusion, top-k, <eos> rule, pruning, finals)
for all U x beam rows; nothing is copied to the host until the search has ended.  `forward_host` is the first
implementation (score table on the host each step), kept as a cross-check."""
import ctypes
import math

import torch
import yaml
from torch import nn

from src import functions as F_hip
from src import hipabi as H
from src.ctc import CTCPrefixScore
from src.lm import RNNLM

CTC_BEAM_RATIO = 1.5
LOG_ZERO = -10000000.0


class Hypothesis(object):
    """History of one partial transcript (reference src/decode.py:186-281); `row` = its row in the device state."""

    def __init__(self, row, output_seq, output_scores, ctc_prob=0.0, ctc_idx=None):
        self.row, self.output_seq, self.output_scores = row, output_seq, output_scores
        self.ctc_prob, self.ctc_idx = ctc_prob, ctc_idx

    def avgScore(self):
        assert len(self.output_scores) != 0
        return sum(self.output_scores) / len(self.output_scores)

    @property
    def outIndex(self):
        return [int(i) for i in self.output_seq]

    def addTopk(self, topi, topv, att_prob, ctc_prob=None, ctc_candidates=None, eos_threshold=1.5):
        new_hyps, term_score = [], None
        for i in range(topi.shape[-1]):
            tok = int(topi[i])
            if tok == 1:
                max_score_no_eos = float(att_prob[2:].max())
                if float(att_prob[tok]) > eos_threshold * max_score_no_eos:
                    term_score = float(topv[i])
                    continue
            cp, ci = None, None
            if ctc_prob is not None:
                ci = ctc_candidates.index(tok)
                cp = float(ctc_prob[ci])
            new_hyps.append(Hypothesis(self.row, self.output_seq + [tok], self.output_scores + [float(topv[i])], cp, ci))
        if term_score is not None:
            self.output_seq.append(1)
            self.output_scores.append(term_score)
            return self, new_hyps
        return None, new_hyps


class BeamDecoder(nn.Module):
    def __init__(self, asr, emb_decoder, beam_size, min_len_ratio, max_len_ratio, lm_path='', lm_config='', lm_weight=0.0,
                 ctc_weight=0.0):
        super().__init__()
        assert emb_decoder is None, 'embedding-fusion decoding is outside the HIP path'
        self.beam_size, self.min_len_ratio, self.max_len_ratio, self.asr = beam_size, min_len_ratio, max_len_ratio, asr
        assert self.asr.enable_att
        if not (self.asr.decoder.fast and self.asr.attention.fast):
            raise NotImplementedError('beam search runs on the decode kernels of the shipped decoder (LSTM, location-aware attention, '
                                      'one head); the model variants of src/variants.py train and validate greedily only')
        self.apply_ctc = ctc_weight > 0
        if self.apply_ctc:
            assert self.asr.ctc_weight > 0, 'ASR was not trained with CTC decoder'
            self.ctc_w = ctc_weight
            self.ctc_beam_size = int(CTC_BEAM_RATIO * self.beam_size)
        self.apply_lm = lm_weight > 0
        if self.apply_lm:
            self.lm_w = lm_weight
            if lm_config:
                cfg = yaml.load(open(lm_config, 'r'), Loader=yaml.FullLoader)
                self.lm = RNNLM(self.asr.vocab_size, **cfg['model'])
                if lm_path:
                    self.lm.load_state_dict(torch.load(lm_path, map_location='cpu')['model'])

    def set_lm(self, lm, weight):
        self.apply_lm, self.lm_w, self.lm = True, weight, lm

    def create_msg(self):
        msg = ['Decode spec| Beam size = {}\t| Min/Max len ratio = {}/{}'.format(self.beam_size, self.min_len_ratio, self.max_len_ratio)]
        if self.apply_ctc:
            msg.append('           |Joint CTC decoding enabled \t| weight = {:.2f}\t'.format(self.ctc_w))
        if self.apply_lm:
            msg.append('           |Joint LM decoding enabled \t| weight = {:.2f}'.format(self.lm_w))
        return msg

    def _encode(self, audio_feature, feature_len):
        """Encoder (+ CTC log-probs) of every utterance on its own, unpadded - the reference decodes one utterance at a
        time (src/decode.py:67) and a padded BiLSTM pass is not equivalent (the reverse direction would start inside the
        padding) - written into zero-padded (U,T'max,.) tensors for the batched search."""
        asr, dev = self.asr, audio_feature.device
        ctx = type('C', (), {'anchor': asr._anchor, 'prec': asr.prec, 'next_seed': lambda s: 0})()
        U = audio_feature.shape[0]
        encs, lens, ctcs = [], [], []
        for u in range(U):
            n = int(feature_len[u])
            e, el = asr.encoder(audio_feature[u:u + 1, :n].float(), feature_len[u:u + 1].to(dev), ctx)
            e = F_hip.to_f32(e)
            encs.append(e[0])
            lens.append(int(el[0]))
            if self.apply_ctc:
                ctcs.append(F_hip.CTCHeadFn.apply(asr._anchor, e, asr.ctc_layer[0], asr.prec, False)[0])
        Tp = max(e.shape[0] for e in encs)
        enc = torch.zeros((U, Tp, encs[0].shape[1]), dtype=torch.float32, device=dev)
        ctc = torch.zeros((U, Tp, asr.vocab_size), dtype=torch.float32, device=dev) if self.apply_ctc else None
        for u in range(U):
            enc[u, :encs[u].shape[0]] = encs[u]
            if ctc is not None:
                ctc[u, :ctcs[u].shape[0]] = ctcs[u]
        # the reference takes T' from the encoder OUTPUT (its masks / CTC scorer see every output frame of the unpadded pass)
        tlen = torch.tensor([e.shape[0] for e in encs], dtype=torch.int32, device=dev)
        return enc, torch.tensor(lens, dtype=torch.int64, device=dev), tlen, ctc

    @torch.no_grad()
    def forward(self, audio_feature, feature_len):
        """audio_feature (U,T,D) zero-padded, feature_len (U).  U == 1: the reference's return value (list of <= beam
        Hypothesis, best first); U > 1: a list of such lists.  No device-to-host copy inside the search loop."""
        asr, dev, st = self.asr, audio_feature.device, H.stream_ptr()
        prec, V, beam = asr.prec, asr.vocab_size, self.beam_size
        U = audio_feature.shape[0]
        flens = [int(x) for x in feature_len.reshape(-1).tolist()]
        max_lens = [int(math.ceil(f * self.max_len_ratio)) for f in flens]
        min_lens = [int(math.ceil(f * self.min_len_ratio)) for f in flens]
        Lmax = max(max(max_lens), 1)
        enc, enc_len, tlen, ctc_lp = self._encode(audio_feature, feature_len)
        Tp, E = enc.shape[1], enc.shape[2]
        R = U * beam
        i32 = lambda *s_: torch.zeros(s_, dtype=torch.int32, device=dev)
        f32 = lambda *s_: torch.zeros(s_, dtype=torch.float32, device=dev)
        d = F_hip._dec_dims(asr, R, Tp, Lmax + 1)
        sd = F_hip._dec_state(d, dev, save_conv=False)
        sd['tokens'].zero_()
        w = H.dec_weights_struct(F_hip._dec_tensors(asr, False), d.NL)
        s = H.dec_state_struct(sd)
        enc_rep = enc.unsqueeze(1).expand(U, beam, Tp, E).reshape(R, Tp, E).contiguous()
        len_rep = enc_len.unsqueeze(1).expand(U, beam).reshape(R).contiguous()
        H.call('asr_att_decoder_keys', ctypes.byref(d), ctypes.byref(w), H.ptr(enc_rep), H.ptr(sd['key']), prec, st)
        C = self.ctc_beam_size if self.apply_ctc else 0
        ctc_r = ctc_rn = cand = psi = None
        if self.apply_ctc:
            ctc_r, ctc_rn = f32(R, Tp, 2), f32(R, C, Tp, 2)
            cand, psi = i32(R, C), f32(R, C)
            H.call('asr_ctc_prefix_init_batched', H.ptr(ctc_lp), H.ptr(tlen), H.ptr(ctc_r), R, beam, Tp, V, st)
        lm_state = None
        if self.apply_lm:
            self.lm.prec = prec
            lm_state = self.lm.init_state(R, dev)
        # hypothesis state, double buffered; row u*beam is the empty hypothesis of utterance u
        state = [{'alive': i32(R), 'sum': f32(R), 'ctcp': f32(R), 'len': i32(R), 'seq': i32(R, Lmax), 'sc': f32(R, Lmax)} for _ in range(2)]
        state[0]['alive'][::beam] = 1
        last_tok, parent, ctc_index = i32(R), torch.zeros(R, dtype=torch.int64, device=dev), torch.zeros(R, dtype=torch.int64, device=dev)
        min_len_d = torch.tensor(min_lens, dtype=torch.int32, device=dev)
        max_len_d = torch.tensor(max_lens, dtype=torch.int32, device=dev)
        done = i32(U)
        fin_n, fin_len, fin_avg = i32(U), i32(U, beam), f32(U, beam)
        fin_seq, fin_sc = i32(U, beam, Lmax + 1), f32(U, beam, Lmax + 1)
        att_lp = f32(R, V)
        tmp = {k: torch.empty_like(sd[k][:, 0]) for k in ('hs', 'cs', 'att')}
        for t in range(Lmax):
            a, b = state[t & 1], state[(t + 1) & 1]
            H.call('asr_att_decoder_step', ctypes.byref(d), ctypes.byref(w), H.ptr(enc_rep), H.ptr(len_rep), ctypes.byref(s), t, prec, st)
            logits = sd['logits'][:, t].contiguous()
            H.call('asr_log_softmax', H.ptr(logits), H.ptr(att_lp), R, V, st)
            if self.apply_ctc:
                H.call('asr_beam_candidates', H.ptr(att_lp), H.ptr(cand), R, V, C, st)
                H.call('asr_ctc_prefix_score_batched', H.ptr(ctc_lp), H.ptr(tlen), H.ptr(ctc_r), H.ptr(cand), H.ptr(a['len']), H.ptr(last_tok),
                       H.ptr(psi), H.ptr(ctc_rn), R, C, Tp, V, beam, st)
            lm_lp = lm_new = None
            if self.apply_lm:
                lm_lp, lm_new = self.lm.step(sd['tokens'][:, t].contiguous(), lm_state)
            args = H.BeamStep()
            for name, ten in (('att_logp', att_lp), ('lm_logp', lm_lp), ('psi', psi), ('candidates', cand), ('alive_in', a['alive']),
                              ('sum_in', a['sum']), ('ctcp_in', a['ctcp']), ('len_in', a['len']), ('seq_in', a['seq']), ('score_in', a['sc']),
                              ('alive_out', b['alive']), ('sum_out', b['sum']), ('ctcp_out', b['ctcp']), ('len_out', b['len']),
                              ('seq_out', b['seq']), ('score_out', b['sc']), ('last_token', last_tok), ('parent', parent),
                              ('ctc_index', ctc_index), ('tokens', sd['tokens']), ('min_len', min_len_d), ('max_len', max_len_d),
                              ('done', done), ('fin_n', fin_n), ('fin_len', fin_len), ('fin_avg', fin_avg), ('fin_seq', fin_seq),
                              ('fin_score', fin_sc)):
                setattr(args, name, ten.data_ptr() if ten is not None else None)
            args.tokens_ld = sd['tokens'].shape[1]
            args.U, args.beam, args.V, args.C, args.Lmax, args.t = U, beam, V, C, Lmax, t
            args.ctc_weight = float(self.ctc_w) if self.apply_ctc else 0.0
            args.lm_weight = float(self.lm_w) if self.apply_lm else 0.0
            args.eos_threshold = 1.5
            H.call('asr_beam_step', ctypes.byref(args), st)
            # state of the survivors: new row i <- row parent[i] (device gathers; no host round trip)
            for name in ('hs', 'cs', 'att'):
                src = sd[name][:, t]
                width = src[0].numel()
                H.call('asr_gather_rows', H.ptr(src), H.ptr(parent), H.ptr(tmp[name]), R, width, src.stride(0), width, R, st)
                src.copy_(tmp[name])
            if self.apply_ctc:
                H.call('asr_gather_rows', H.ptr(ctc_rn), H.ptr(ctc_index), H.ptr(ctc_r), R, Tp * 2, Tp * 2, Tp * 2, R * C, st)
            if self.apply_lm:
                hn, cn = torch.empty_like(lm_new[0]), torch.empty_like(lm_new[1])
                for l in range(self.lm.n_layers):
                    H.call('asr_gather_rows', H.ptr(lm_new[0][l]), H.ptr(parent), H.ptr(hn[l]), R, self.lm.dim, self.lm.dim, self.lm.dim, R, st)
                    H.call('asr_gather_rows', H.ptr(lm_new[1][l]), H.ptr(parent), H.ptr(cn[l]), R, self.lm.dim, self.lm.dim, self.lm.dim, R, st)
                lm_state = (hn, cn)
            if (t & 15) == 15 and bool(done.all()):          # every 16 positions: stop early when every search has ended
                break
        n_c, len_c, seq_c, sc_c = fin_n.cpu(), fin_len.cpu(), fin_seq.cpu(), fin_sc.cpu()
        out = []
        for u in range(U):
            hyps = []
            for i in range(int(n_c[u])):
                l = int(len_c[u, i])
                hyps.append(Hypothesis(i, seq_c[u, i, :l].tolist(), sc_c[u, i, :l].tolist()))
            out.append(hyps)
        return out[0] if U == 1 else out

    @torch.no_grad()
    def forward_host(self, audio_feature, feature_len):
        assert audio_feature.shape[0] == 1, 'Batchsize == 1 is required for beam search'
        asr, dev, st = self.asr, audio_feature.device, H.stream_ptr()
        prec = asr.prec
        flen = int(feature_len.reshape(-1)[0])
        max_len = int(math.ceil(flen * self.max_len_ratio))
        min_len = int(math.ceil(flen * self.min_len_ratio))
        ctx = type('C', (), {'anchor': asr._anchor, 'prec': prec, 'next_seed': lambda s: 0})()
        enc, enc_len = asr.encoder(audio_feature.float(), feature_len.to(dev), ctx)
        enc = F_hip.to_f32(enc)
        Tp = enc.shape[1]
        nmax = self.beam_size
        L = max(max_len, 1)
        d = F_hip._dec_dims(asr, nmax, Tp, L)
        sd = F_hip._dec_state(d, dev, save_conv=False)
        w = H.dec_weights_struct(F_hip._dec_tensors(asr, False), d.NL)
        s = H.dec_state_struct(sd)
        enc_rep = enc.expand(nmax, Tp, enc.shape[2]).contiguous()
        len_rep = enc_len.to(dev, torch.int64).expand(nmax).contiguous()
        H.call('asr_att_decoder_keys', ctypes.byref(d), ctypes.byref(w), H.ptr(enc_rep), H.ptr(sd['key']), prec, st)
        ctc_r = None
        if self.apply_ctc:
            ctc_lp = F_hip.CTCHeadFn.apply(asr._anchor, enc, asr.ctc_layer[0], prec, False)
            scorer = CTCPrefixScore(ctc_lp)
            ctc_r = scorer.init_state().unsqueeze(0).repeat(nmax, 1, 1).contiguous()
        lm_state = self.lm.init_state(nmax, dev) if self.apply_lm else None
        if self.apply_lm:
            self.lm.prec = prec
        V = asr.vocab_size
        hyps = [Hypothesis(0, [], [], 0.0)]
        finals = []
        for t in range(max_len):
            n = len(hyps)
            toks = torch.tensor([h.output_seq[-1] if h.output_seq else 0 for h in hyps], dtype=torch.int64)
            sd['tokens'][:n, t] = toks.to(dev)
            d.B = n
            H.call('asr_att_decoder_step', ctypes.byref(d), ctypes.byref(w), H.ptr(enc_rep), H.ptr(len_rep), ctypes.byref(s), t, prec, st)
            logits = sd['logits'][:n, t].contiguous()
            att_lp = torch.empty_like(logits)
            H.call('asr_log_softmax', H.ptr(logits), H.ptr(att_lp), n, V, st)
            att_cpu = att_lp.cpu()
            cur = att_cpu.clone()
            cands, psi_cpu, r_new = None, None, None
            if self.apply_ctc:
                _, cand = att_cpu.topk(self.ctc_beam_size, dim=-1)
                psi, r_new = scorer.score([len(h.output_seq) for h in hyps], [h.output_seq[-1] if h.output_seq else 0 for h in hyps],
                                          ctc_r[:n], cand)
                psi_cpu, cands = psi.cpu(), cand.tolist()
                for i, h in enumerate(hyps):
                    hack = torch.full((V,), LOG_ZERO)
                    hack[cand[i]] = psi_cpu[i] - float(h.ctc_prob)
                    cur[i] = (1 - self.ctc_w) * cur[i] + self.ctc_w * hack
                    cur[i, 0] = LOG_ZERO
            if self.apply_lm:
                lm_lp, lm_new = self.lm.step(toks.to(dev), (lm_state[0][:, :n].contiguous(), lm_state[1][:, :n].contiguous()))
                cur += self.lm_w * lm_lp.cpu()
            children = []
            for i, h in enumerate(hyps):
                h.row = i
                topv, topi = cur[i].topk(self.beam_size)
                final, new = h.addTopk(topi, topv, att_cpu[i], None if psi_cpu is None else psi_cpu[i], None if cands is None else cands[i])
                if final is not None and t >= min_len:
                    finals.append(final)
                    if self.beam_size == 1:
                        return finals
                children.extend(new)
            children.sort(key=lambda o: o.avgScore(), reverse=True)
            hyps = children[:self.beam_size]
            if not hyps:
                break
            # device state of the survivors: row i <- row parent(i) at step t
            par = torch.tensor([h.row for h in hyps], dtype=torch.int64, device=dev)
            m = len(hyps)
            for name in ('hs', 'cs', 'att'):
                sd[name][:m, t] = sd[name][par, t]
            if self.apply_ctc:
                ci = torch.tensor([h.ctc_idx for h in hyps], dtype=torch.int64, device=dev)
                ctc_r[:m] = r_new[par, ci]
            if self.apply_lm:
                lm_state[0][:, :m] = lm_new[0][:, par]
                lm_state[1][:, :m] = lm_new[1][:, par]
        finals += hyps
        finals.sort(key=lambda o: o.avgScore(), reverse=True)
        return finals[:self.beam_size]
