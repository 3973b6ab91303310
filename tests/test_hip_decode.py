"""Beam-search decoding on the HIP path vs the reference fixtures (tests/golden/g7_decode.npz): CTC prefix scores,
and the hypotheses (token sequences + per-token scores) for attention-only, +CTC and +CTC+RNNLM decoding."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def g7(golden_dir):
    z = np.load(os.path.join(golden_dir, 'g7_decode.npz'))
    return yaml.safe_load(str(z['meta'])), z


def test_ctc_prefix_scores_match_reference(g7):
    from src.ctc import CTCPrefixScore
    meta, z = g7
    lp = torch.from_numpy(z['ctc_logp']).unsqueeze(0).cuda()
    ps = CTCPrefixScore(lp)
    r0 = ps.init_state()
    np.testing.assert_allclose(r0.cpu().numpy(), z['r0'], rtol=1e-5, atol=1e-4)
    cand = z['cand'].tolist()
    psi1, r1 = ps.cheap_compute([], r0, cand)
    np.testing.assert_allclose(psi1.cpu().numpy(), z['psi1'], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(r1.cpu().numpy(), z['r1'], rtol=2e-5, atol=2e-3)
    psi2, r2 = ps.cheap_compute([7], r1[cand.index(7)], cand)
    np.testing.assert_allclose(psi2.cpu().numpy(), z['psi2'], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(r2.cpu().numpy(), z['r2'], rtol=2e-5, atol=2e-3)
    psi3, r3 = ps.cheap_compute([7, 7], r2[cand.index(7)], [7, 1, 3])
    np.testing.assert_allclose(psi3.cpu().numpy(), z['psi3'], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize('tag,ctc_w,lm_w', [('att', 0.0, 0.0), ('ctc', 0.3, 0.0), ('ctc_lm', 0.3, 0.5)])
def test_beam_hypotheses_match_reference(g7, tag, ctc_w, lm_w):
    from src.asr import ASR
    from src.decode import BeamDecoder
    from src.lm import RNNLM
    meta, z = g7
    cfg = O.ModelCfg(meta['model'], meta['D'], meta['V'])
    model = ASR(meta['D'], meta['V'], 4, prec='fp32', **meta['model'])
    model.load_state_dict(O.seeded_state_dict(O.param_shapes(cfg), meta['wseed']))
    model = model.cuda().eval()
    dec = BeamDecoder(model, None, beam_size=meta['beam'], min_len_ratio=meta['min_len_ratio'], max_len_ratio=meta['max_len_ratio'],
                      ctc_weight=ctc_w)
    if lm_w > 0:
        lm = RNNLM(meta['V'], **meta['lm'])
        shapes = {k: tuple(v.shape) for k, v in lm.state_dict().items()}
        lm.load_state_dict(O.seeded_state_dict(shapes, meta['lm_wseed']))
        dec.set_lm(lm.cuda().eval(), lm_w)
    hyps = dec(torch.from_numpy(z['feat']).cuda(), torch.from_numpy(z['feat_len']).cuda())
    assert len(hyps) == int(z['n_' + tag])
    for i, h in enumerate(hyps):
        assert h.outIndex == z['%s_seq%d' % (tag, i)].tolist(), (tag, i, h.outIndex, z['%s_seq%d' % (tag, i)].tolist())
        np.testing.assert_allclose(np.array(h.output_scores, dtype=np.float32), z['%s_score%d' % (tag, i)], rtol=1e-4, atol=2e-3)
        assert abs(h.avgScore() - float(z['%s_avg%d' % (tag, i)])) < 2e-3


def _decoder(meta, ctc_w, lm_w, beam=None):
    from src.asr import ASR
    from src.decode import BeamDecoder
    from src.lm import RNNLM
    cfg = O.ModelCfg(meta['model'], meta['D'], meta['V'])
    model = ASR(meta['D'], meta['V'], 4, prec='fp32', **meta['model'])
    model.load_state_dict(O.seeded_state_dict(O.param_shapes(cfg), meta['wseed']))
    model = model.cuda().eval()
    dec = BeamDecoder(model, None, beam_size=beam or meta['beam'], min_len_ratio=meta['min_len_ratio'], max_len_ratio=meta['max_len_ratio'],
                      ctc_weight=ctc_w)
    if lm_w > 0:
        lm = RNNLM(meta['V'], **meta['lm'])
        shapes = {k: tuple(v.shape) for k, v in lm.state_dict().items()}
        lm.load_state_dict(O.seeded_state_dict(shapes, meta['lm_wseed']))
        dec.set_lm(lm.cuda().eval(), lm_w)
    return dec


@pytest.mark.parametrize('ctc_w,lm_w,beam', [(0.0, 0.0, 4), (0.3, 0.5, 4), (0.3, 0.5, 8), (0.0, 0.0, 1), (0.3, 0.0, 1)])
def test_device_beam_search_batched_equals_single_and_host_path(g7, ctc_w, lm_w, beam):
    """The search that stays on the device (asr_beam_step) against (i) the first implementation with the score table on
    the host and (ii) itself on a zero-padded BATCH of utterances of different lengths: every utterance must come out
    exactly as when decoded alone (the reference decodes one utterance at a time)."""
    meta, z = g7
    dec = _decoder(meta, ctc_w, lm_w, beam)
    g = np.random.Generator(np.random.PCG64(7))
    lens = [61, 40, 53]
    feats = torch.zeros(3, 61, meta['D'])
    feats[0] = torch.from_numpy(z['feat'][0])
    for u in (1, 2):
        feats[u, :lens[u]] = torch.from_numpy(g.random((lens[u], meta['D']), dtype=np.float32))
    flen = torch.tensor(lens)
    batched = dec(feats.cuda(), flen.cuda())
    assert len(batched) == 3
    for u in range(3):
        single = dec(feats[u:u + 1, :lens[u]].cuda(), flen[u:u + 1].cuda())
        host = dec.forward_host(feats[u:u + 1, :lens[u]].cuda(), flen[u:u + 1].cuda())
        assert len(single) == len(batched[u]) == len(host) > 0, (u, len(single), len(batched[u]), len(host))
        for a, b, c in zip(single, batched[u], host):
            assert a.outIndex == b.outIndex == c.outIndex, (u, a.outIndex, b.outIndex, c.outIndex)
            np.testing.assert_allclose(np.array(a.output_scores), np.array(b.output_scores), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(np.array(a.output_scores), np.array(c.output_scores), rtol=1e-4, atol=2e-3)


def test_config4_size_beam_search_matches_reference(golden_dir):
    """BASELINE config 4 at its size: the 12 M-parameter model of config/librispeech_asr.yaml, beam 8,
    ctc_weight 0.3, the 4 x 1024 tied RNN-LM with weight 0.3, T = 400 / 363 / 326 frames, max_len_ratio 0.05 - against
    hypotheses and scores the imported reference produced (tests/golden/gen_golden.py::gen_decode_config4), decoded one utterance
    at a time and as one zero-padded batch.  With seeded random weights the reference's eight hypotheses are near-ties (average
    scores within 3e-3 of each other), so the comparison is by content, not by rank: the best average score, every reference
    hypothesis that we also return token for token with its per-token scores, at least six of the eight sequences in common,
    and nothing of ours worse than the reference's worst by more than the tie margin.  fp32 contraction mode."""
    from src.asr import ASR
    from src.decode import BeamDecoder
    from src.lm import RNNLM
    z = np.load(os.path.join(golden_dir, 'g7b_decode_config4.npz'))
    meta = yaml.safe_load(str(z['meta']))
    cfg = O.ModelCfg(meta['model'], meta['D'], meta['V'])
    model = ASR(meta['D'], meta['V'], 8, prec='fp32', **meta['model'])
    model.load_state_dict(O.seeded_state_dict(O.param_shapes(cfg), meta['wseed']))
    model = model.cuda().eval()
    dec = BeamDecoder(model, None, beam_size=meta['beam'], min_len_ratio=meta['min_len_ratio'], max_len_ratio=meta['max_len_ratio'],
                      ctc_weight=meta['ctc_weight'])
    lm = RNNLM(meta['V'], **meta['lm'])
    lm.load_state_dict(O.seeded_state_dict({k: tuple(v.shape) for k, v in lm.state_dict().items()}, meta['lm_wseed']))
    dec.set_lm(lm.cuda().eval(), meta['lm_weight'])
    nutt = meta['nutt']
    lens = [int(z['feat_len%d' % u][0]) for u in range(nutt)]
    feats = torch.zeros(nutt, max(lens), meta['D'])
    for u in range(nutt):
        feats[u, :lens[u]] = torch.from_numpy(z['feat%d' % u][0])
    batched = dec(feats.cuda(), torch.tensor(lens).cuda())

    def check(hyps, u, what):
        ref = [(z['u%d_seq%d' % (u, i)].tolist(), z['u%d_score%d' % (u, i)], float(z['u%d_avg%d' % (u, i)])) for i in range(int(z['n%d' % u]))]
        assert len(hyps) == len(ref), (what, u, len(hyps), len(ref))
        assert abs(hyps[0].avgScore() - ref[0][2]) < 2e-3, (what, u, hyps[0].avgScore(), ref[0][2])
        ours = {tuple(h.outIndex): h for h in hyps}
        common = 0
        for seq, scores, avg in ref:
            h = ours.get(tuple(seq))
            if h is not None:
                common += 1
                np.testing.assert_allclose(np.array(h.output_scores, dtype=np.float32), scores, rtol=1e-4, atol=2e-3)
                assert abs(h.avgScore() - avg) < 2e-3
        assert common >= len(ref) - 2, (what, u, common, [h.outIndex for h in hyps], [r[0] for r in ref])
        worst = min(r[2] for r in ref)
        assert all(h.avgScore() > worst - 3e-3 for h in hyps), (what, u)
        assert all(len(h.outIndex) == len(ref[0][0]) for h in hyps)

    for u in range(nutt):
        check(batched[u], u, 'batched')
        check(dec(feats[u:u + 1, :lens[u]].cuda(), torch.tensor(lens[u:u + 1]).cuda()), u, 'single')


@pytest.mark.parametrize('seed,T,beam,ctc_w,lm_w', [(301, 47, 3, 0.4, 0.0), (302, 83, 5, 0.3, 0.6), (303, 29, 2, 0.0, 0.4), (304, 120, 6, 0.5, 0.2)])
def test_beam_search_matches_decode_oracle_on_new_inputs(g7, seed, T, beam, ctc_w, lm_w):
    """Seeded utterances that are NOT in the fixtures, other beam widths and fusion weights: the HIP decoder against the CPU
    restatement of src/decode.py / src/ctc.py / src/lm.py (oracle/decode_oracle.py, itself pinned to the reference's hypotheses by
    tests/test_oracle_golden.py::test_decode_*).  Token sequences must be identical; scores within fp32 accumulation noise."""
    from oracle import decode_oracle as D
    meta, _ = g7
    dec = _decoder(meta, ctc_w, lm_w, beam=beam)
    dec.max_len_ratio = 0.15
    g = np.random.Generator(np.random.PCG64(seed))
    feat = torch.from_numpy(g.random((1, T, meta['D']), dtype=np.float32))
    flen = torch.tensor([T], dtype=torch.int64)
    hyps = dec(feat.cuda(), flen.cuda())
    cfg = O.ModelCfg(meta['model'], meta['D'], meta['V'])
    P = O.seeded_state_dict(O.param_shapes(cfg), meta['wseed'])
    lm = None
    if lm_w > 0:
        P_lm = {k: v.detach().cpu() for k, v in dec.lm.state_dict().items()}
        lm = (P_lm, meta['lm'])
    ref = D.beam_search(feat, flen, P, cfg, beam, meta['min_len_ratio'], 0.15, ctc_weight=ctc_w, lm=lm, lm_weight=lm_w)
    assert len(hyps) == len(ref)
    for i, (h, (seq, scores)) in enumerate(zip(hyps, ref)):
        assert h.outIndex == seq, (i, h.outIndex, seq)
        np.testing.assert_allclose(np.array(h.output_scores, dtype=np.float32), np.array(scores, np.float32), rtol=1e-4, atol=2e-3)
