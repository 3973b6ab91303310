"""Beam-search decoder with the reference's behaviour (src/decode.py:65-281) — hypothesis expansion, the <eos>
threshold rule, joint CTC / RNN-LM scoring, average-score pruning — but every model evaluation is batched over the
live hypotheses on the GPU: one attention+decoder step, one CTC prefix-score launch and one LM step per output
position instead of one per hypothesis.  Only the (beam x V) score table visits the host for the top-k book-keeping."""
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

    @torch.no_grad()
    def forward(self, audio_feature, feature_len):
        assert audio_feature.shape[0] == 1, 'Batchsize == 1 is required for beam search'
        asr, dev, st = self.asr, audio_feature.device, H.stream_ptr()
        prec = asr.prec
        flen = int(feature_len.reshape(-1)[0])
        max_len = int(math.ceil(flen * self.max_len_ratio))
        min_len = int(math.ceil(flen * self.min_len_ratio))
        ctx = type('C', (), {'anchor': asr._anchor, 'prec': prec, 'next_seed': lambda s: 0})()
        enc, enc_len = asr.encoder(audio_feature.float(), feature_len.to(dev), ctx)
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
