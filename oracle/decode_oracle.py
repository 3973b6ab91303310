"""CPU restatement of the reference's INFERENCE leg - TEST INFRASTRUCTURE ONLY (imported by tests/ alone; the product path is
e2e-asr-pytorch_amd/src/decode.py on the decode kernels of csrc/decode.hip and never touches this file).

  ctc_prefix_init / ctc_prefix_cheap : CTCPrefixScore.init_state / cheap_compute       reference src/ctc.py:19-27, 68-107
  rnnlm_step                         : RNNLM.forward for one token (eval mode)         reference src/lm.py:27-38
  beam_search                        : BeamDecoder.forward + Hypothesis.addTopk         reference src/decode.py:65-183, 214-263
The acoustic model pieces (encoder, CTC head, attention step, decoder cell) are the ones of oracle/asr_oracle.py.

Pinned to the genuine reference by tests/test_oracle_golden.py::test_decode_* against tests/golden/g7_decode.npz (beam 4: attention
only, + CTC 0.3, + CTC 0.3 + LM 0.5; prefix-scorer states) and g7b_decode_config4.npz (BASELINE config 4 at its size: the 12 M
parameter model, beam 8, CTC 0.3, 4 x 1024 LM 0.3, three utterances) - hypotheses, per-token scores and averages.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import asr_oracle as O

LOGZERO_CTC = -100000000.0      # src/ctc.py:12
LOG_ZERO = -10000000.0          # src/decode.py:11
CTC_BEAM_RATIO = 1.5            # src/decode.py:10


# ---- CTC prefix scores (src/ctc.py) ----------------------------------------------------------------------------------
def ctc_prefix_init(x):
    """x (T, V) log-probs -> r (T, 2): r[:, 0] = log zero (non-blank), r[t, 1] = cumulative blank log-prob."""
    T = x.shape[0]
    r = np.full((T, 2), LOGZERO_CTC, dtype=np.float32)
    r[0, 1] = x[0, 0]
    for i in range(1, T):
        r[i, 1] = r[i - 1, 1] + x[i, 0]
    return r


def ctc_prefix_cheap(x, g, r_prev, candidates):
    """Prefix g (list of ints), previous state r_prev (T, 2), candidate tokens -> (psi (C,), r (C, T, 2)).  float32 numpy as the
    reference computes it (np.logaddexp on float32 arrays)."""
    T = x.shape[0]
    C = len(candidates)
    last = g[-1] if len(g) > 0 else 0
    r = np.full((T, 2, C), LOGZERO_CTC, dtype=np.float32)
    start = max(1, len(g))
    if len(g) == 0:
        r[0, 0, :] = x[0, candidates]
    psi = r[start - 1, 0, :]
    sum_prev = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
    phi = np.repeat(sum_prev[..., None], C, axis=-1)
    if len(g) > 0 and last in candidates:
        phi[:, candidates.index(last)] = r_prev[:, 1]
    for t in range(start, T):
        r[t, 0, :] = np.logaddexp(r[t - 1, 0, :], phi[t - 1]) + x[t, candidates]
        r[t, 1, :] = np.logaddexp(r[t - 1, 1, :], r[t - 1, 0, :]) + x[t, 0]
        psi = np.logaddexp(psi, phi[t - 1, ] + x[t, candidates])
    if 1 in candidates:
        psi[candidates.index(1)] = sum_prev[-1]
    return psi, np.rollaxis(r, 2)


# ---- RNN-LM step (src/lm.py) ----------------------------------------------------------------------------------------------
def rnnlm_step(P, cfg, token, state):
    """One token through emb -> n_layers LSTM -> (tied) projection.  state: None or (h [n_layers x (1,dim)], c [...])."""
    nl, dim = int(cfg['n_layers']), int(cfg['dim'])
    x = P['emb.weight'][token].view(1, -1)
    if state is None:
        state = ([torch.zeros(1, dim) for _ in range(nl)], [torch.zeros(1, dim) for _ in range(nl)])
    h, c = [t.clone() for t in state[0]], [t.clone() for t in state[1]]
    for l in range(nl):
        h[l], c[l] = O.lstm_cell(x, h[l], c[l], P['rnn.weight_ih_l%d' % l], P['rnn.weight_hh_l%d' % l], P['rnn.bias_ih_l%d' % l],
                                 P['rnn.bias_hh_l%d' % l])
        x = h[l]
    out = x @ P['emb.weight'].t() if cfg['emb_tying'] else x @ P['trans.weight'].t() + P['trans.bias']
    return out, (h, c)


# ---- beam search (src/decode.py) --------------------------------------------------------------------------------------
class Hyp(object):
    def __init__(self, dec_state, seq, scores, lm_state, ctc_state, ctc_prob, att_map):
        self.dec_state, self.seq, self.scores, self.lm_state, self.ctc_state, self.ctc_prob, self.att_map = \
            dec_state, seq, scores, lm_state, ctc_state, ctc_prob, att_map

    def avg(self):
        return sum(self.scores) / len(self.scores)


def beam_search(feat, feat_len, P, cfg, beam_size, min_len_ratio, max_len_ratio, ctc_weight=0.0, lm=None, lm_weight=0.0,
                eos_threshold=1.5):
    """feat (1, T, D), feat_len (1,).  lm: None or (P_lm, lm_cfg).  Returns the reference's list: [(tokens, scores)] best first."""
    with torch.no_grad():
        enc, enc_len = O.encoder(feat, feat_len, P, cfg)
        Tp = enc.shape[1]
        mask = (torch.arange(Tp)[None, :] >= enc_len[:, None])
        key = O.attention_keys(enc, P)
        max_len = int(np.ceil(int(feat_len[0]) * max_len_ratio))
        min_len = int(np.ceil(int(feat_len[0]) * min_len_ratio))
        x_ctc, ctc_state0, ctc_beam = None, None, 0
        if ctc_weight > 0:
            x_ctc = O.ctc_head(enc, P)[0].numpy()
            ctc_state0 = ctc_prefix_init(x_ctc)
            ctc_beam = int(CTC_BEAM_RATIO * beam_size)
        dim, nl = cfg.dec_dim, cfg.dec_layer
        zero = ([torch.zeros(1, dim) for _ in range(nl)], [torch.zeros(1, dim) for _ in range(nl)])
        uniform = torch.where(mask, torch.zeros(()), (1.0 / enc_len.float())[:, None].expand(1, Tp))
        prev_top = [Hyp(zero, [], [], None, ctc_state0, 0, None)]
        finals, nxt = [], []
        E = P['pre_embed.weight']
        for t in range(max_len):
            for hyp in prev_top:
                tok = hyp.seq[-1] if len(hyp.seq) else 0
                h, c = hyp.dec_state
                prev_att = hyp.att_map if hyp.att_map is not None else uniform
                attn, ctx = O.loc_attention_step(torch.cat(h, dim=-1), key, enc, prev_att, mask, P, cfg)
                x = torch.cat([E[torch.tensor([tok])], ctx], dim=-1)
                h2, c2 = [], []
                for l in range(nl):
                    hl, cl = O.lstm_cell(x, h[l], c[l], P['decoder.layers.weight_ih_l%d' % l], P['decoder.layers.weight_hh_l%d' % l],
                                         P['decoder.layers.bias_ih_l%d' % l], P['decoder.layers.bias_hh_l%d' % l])
                    h2.append(hl); c2.append(cl)
                    x = hl
                cur = F.log_softmax(x @ P['decoder.char_trans.weight'].t() + P['decoder.char_trans.bias'], dim=-1)      # (1, V)
                att_prob = cur[0].clone()
                ctc_state, ctc_prob, cand = None, None, None
                if ctc_weight > 0:
                    cand = cur[0].topk(ctc_beam)[1].tolist()
                    ctc_prob, ctc_state = ctc_prefix_cheap(x_ctc, hyp.seq, hyp.ctc_state, cand)
                    ctc_char = torch.from_numpy(np.asarray(ctc_prob - hyp.ctc_prob, dtype=np.float32))
                    hack = torch.full_like(cur, LOG_ZERO)
                    for i, ch in enumerate(cand):
                        hack[0, ch] = ctc_char[i]
                    cur = (1 - ctc_weight) * cur + ctc_weight * hack
                    cur[0, 0] = LOG_ZERO
                lm_state = None
                if lm is not None and lm_weight > 0:
                    lm_out, lm_state = rnnlm_step(lm[0], lm[1], tok, hyp.lm_state)
                    cur = cur + lm_weight * F.log_softmax(lm_out, dim=-1)
                topv, topi = cur[0].topk(beam_size)
                # Hypothesis.addTopk (src/decode.py:214-263)
                new, term = [], None
                for i in range(beam_size):
                    ti = int(topi[i])
                    if ti == 1:
                        if float(att_prob[1]) > eos_threshold * float(att_prob[2:].max()):
                            term = float(topv[i])
                            continue
                    cs, cp = None, None
                    if ctc_state is not None:
                        j = cand.index(ti)
                        cs, cp = ctc_state[j], ctc_prob[j]
                    new.append(Hyp((h2, c2), hyp.seq + [ti], hyp.scores + [float(topv[i])], lm_state, cs, cp, attn))
                if term is not None:
                    hyp.seq = hyp.seq + [1]
                    hyp.scores = hyp.scores + [term]
                    if t >= min_len:
                        finals.append(hyp)
                        if beam_size == 1:
                            return [(hyp.seq, hyp.scores)]
                nxt.extend(new)
            nxt.sort(key=lambda o: o.avg(), reverse=True)          # stable, like list.sort in the reference
            prev_top, nxt = nxt[:beam_size], []
        finals += prev_top
        finals.sort(key=lambda o: o.avg(), reverse=True)
        return [(hh.seq, hh.scores) for hh in finals[:beam_size]]
