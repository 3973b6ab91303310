"""CTC prefix scorer for joint decoding (reference src/ctc.py:4-108, Watanabe et al. Algo. 2) on the HIP path:
states live on the device as (N,T,2) fp32, `score` evaluates N hypotheses x C candidates in one launch."""
import torch

from src import hipabi as H


class CTCPrefixScore(object):
    def __init__(self, x):
        """x: (1,T,V) CTC log-probs of one utterance (device tensor)."""
        self.logzero, self.blank, self.eos = -100000000.0, 0, 1
        self.x = x[0].contiguous().float()
        self.input_length, self.odim = self.x.shape

    def init_state(self):
        r = torch.empty((self.input_length, 2), dtype=torch.float32, device=self.x.device)
        H.call('asr_ctc_prefix_init', H.ptr(self.x), H.ptr(r), self.input_length, self.odim, H.stream_ptr())
        return r

    def score(self, prefix_len, last_token, r_prev, candidates):
        """prefix_len, last_token: (N) ints; r_prev (N,T,2); candidates (N,C) -> psi (N,C), r_new (N,C,T,2)."""
        dev = self.x.device
        N, C = candidates.shape
        cand = candidates.to(dev, torch.int32).contiguous()
        pl = torch.as_tensor(prefix_len, dtype=torch.int32).to(dev)
        lt = torch.as_tensor(last_token, dtype=torch.int32).to(dev)
        r_prev = r_prev.contiguous()
        psi = torch.empty((N, C), dtype=torch.float32, device=dev)
        r_new = torch.empty((N, C, self.input_length, 2), dtype=torch.float32, device=dev)
        H.call('asr_ctc_prefix_score', H.ptr(self.x), H.ptr(r_prev), H.ptr(cand), H.ptr(pl), H.ptr(lt), H.ptr(psi), H.ptr(r_new),
               N, C, self.input_length, self.odim, H.stream_ptr())
        return psi, r_new

    def cheap_compute(self, g, r_prev, candidates):
        """Single-hypothesis form with the reference's signature (returns device tensors)."""
        psi, r = self.score([len(g)], [g[-1] if len(g) > 0 else 0], r_prev.unsqueeze(0),
                            torch.as_tensor(candidates).view(1, -1))
        return psi[0], r[0]
