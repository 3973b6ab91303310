"""Pins the CPU oracle (oracle/asr_oracle.py) to the fixtures generated from the genuine reference
(tests/golden/gen_golden.py).  CPU only; runs in the build container and on the GPU box alike."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import asr_oracle as O

torch.set_num_threads(4)


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)
    meta = yaml.safe_load(str(z['meta']))
    return meta, z


def run_oracle(meta, z, feat=None, drop_masks=None, lstm_impl=O.bilstm, dec_masks=None):
    cfg = O.ModelCfg(meta['model'], meta['D'], meta['V'])
    P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_state_dict(O.param_shapes(cfg), meta['wseed']).items()}
    feat = torch.from_numpy(z['feat']) if feat is None else feat
    res = O.asr_losses(feat, torch.from_numpy(z['feat_len']), torch.from_numpy(z['txt']), P, cfg,
                       label_smoothing=meta['label_smoothing'], drop_masks=drop_masks, lstm_impl=lstm_impl, dec_masks=dec_masks)
    res['total_loss'].backward()
    return cfg, P, res


SMALL = ['g1_small_c2', 'g1_small_c2_dropout', 'g3_small_debug', 'g3_small_ln_concat']


@pytest.mark.parametrize('name', SMALL)
@pytest.mark.parametrize('impl', ['loop', 'aten'])
def test_small_models_match_reference(golden_dir, name, impl):
    meta, z = load(golden_dir, name)
    masks = None
    if 'mask0' in z.files:
        masks = [torch.from_numpy(z['mask0']), torch.from_numpy(z['mask1'])]
    cfg, P, res = run_oracle(meta, z, drop_masks=masks, lstm_impl=O.bilstm if impl == 'loop' else O.bilstm_aten)
    assert np.array_equal(res['enc_len'].numpy(), z['enc_len'])
    for key in ('ctc_output', 'att_output', 'att_seq'):
        if key in z.files:
            np.testing.assert_allclose(res[key].detach().numpy(), z[key], atol=2e-5, rtol=1e-5)
    for key in ('ctc_loss', 'att_loss', 'total_loss'):
        if key in z.files:
            assert abs(float(res[key].detach()) - float(z[key])) < 1e-5 * max(1.0, abs(float(z[key])))
    for k, p in P.items():
        g = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        ref = z['grad.' + k]
        np.testing.assert_allclose(g, ref, atol=3e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)
    total, scale = O.clip_grad_norm([p.grad for p in P.values() if p.grad is not None])
    assert abs(total - float(z['grad_norm'])) < 1e-4 * max(1.0, total)


VARIANTS = ['g10_dot', 'g10_loc_mh_vproj', 'g10_dot_mh', 'g10_gru', 'g10_decdrop', 'g10_vgg7']


def variant_masks(z):
    if 'mask_emb' not in z.files:
        return None
    return {'emb': torch.from_numpy(z['mask_emb']), 'final': [torch.from_numpy(m) for m in z['mask_final']], 'layer': None}


@pytest.mark.parametrize('name', VARIANTS)
def test_variant_models_match_reference(golden_dir, name):
    """scaled-dot / multi-head / value-projected attention, GRU encoder + 2-layer GRU decoder, decoder and
    embedding dropout with known masks - the oracle's restatement against the genuine reference's outputs and gradients."""
    meta, z = load(golden_dir, name)
    cfg, P, res = run_oracle(meta, z, dec_masks=variant_masks(z))
    assert np.array_equal(res['enc_len'].numpy(), z['enc_len'])
    for key in ('ctc_output', 'att_output', 'att_seq'):
        if key in z.files:
            np.testing.assert_allclose(res[key].detach().numpy(), z[key], atol=2e-5, rtol=1e-5, err_msg=key)
    for key in ('ctc_loss', 'att_loss', 'total_loss'):
        if key in z.files:
            assert abs(float(res[key].detach()) - float(z[key])) < 1e-5 * max(1.0, abs(float(z[key])))
    for k, p in P.items():
        g = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        ref = z['grad.' + k]
        np.testing.assert_allclose(g, ref, atol=3e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


@pytest.mark.parametrize('name', ['g3_small_vgg1', 'g3_small_vgg5', 'g10_vgg2', 'g10_vgg3', 'g10_vgg4'])
def test_vgg_models_match_reference(golden_dir, name):
    meta, z = load(golden_dir, name)
    cfg, P, res = run_oracle(meta, z)
    assert np.array_equal(res['enc_len'].numpy(), z['enc_len'])
    np.testing.assert_allclose(res['ctc_output'].detach().numpy(), z['ctc_output'], atol=5e-5)
    np.testing.assert_allclose(res['att_output'].detach().numpy(), z['att_output'], atol=5e-5)
    for k, p in P.items():
        ref = float(z['gradnorm.' + k])
        assert abs(float(p.grad.norm()) - ref) < 1e-4 * max(ref, 1e-3) + 1e-6, k
        np.testing.assert_allclose(p.grad.reshape(-1)[:8].numpy(), z['gradhead.' + k], atol=1e-5 + 1e-4 * ref)


def test_full_size_config_matches_reference(golden_dir):
    """config/librispeech_asr.yaml model (12.08 M parameters), B=4, T=203, L=11 — summaries."""
    meta, z = load(golden_dir, 'g2_full_c2')
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_golden_shapes', os.path.join(golden_dir, 'batchgen.py'))
    bg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bg)
    feat, lens, txt = bg.make_batch(int(z['feat_seed']), 4, 203, 160, 11, 31)
    assert np.array_equal(lens, z['feat_len']) and np.array_equal(txt, z['txt'])
    cfg, P, res = run_oracle(meta, z, feat=torch.from_numpy(feat), lstm_impl=O.bilstm_aten)
    assert sum(int(np.prod(p.shape)) for p in P.values()) == 12079853
    for key in ('ctc_loss', 'att_loss', 'total_loss'):
        assert abs(float(res[key].detach()) - float(z[key])) < 2e-5 * max(1.0, abs(float(z[key])))
    np.testing.assert_allclose(res['ctc_output'].detach().numpy()[:, :4], z['ctc_output_head'], atol=5e-5)
    np.testing.assert_allclose(res['att_output'].detach().numpy()[:, :4], z['att_output_head'], atol=5e-5)
    np.testing.assert_allclose(res['att_seq'].detach().numpy()[:, :, :4], z['att_seq_head'], atol=1e-5)
    for k, p in P.items():
        ref = float(z['gradnorm.' + k])
        assert abs(float(p.grad.norm()) - ref) < 2e-4 * max(ref, 1e-3) + 1e-6, (k, float(p.grad.norm()), ref)


def test_ctc_restatement_matches_torch_and_bruteforce(golden_dir):
    _, z = load(golden_dir, 'g4_ctc')
    logits, txt, in_len = z['logits'], z['txt'], z['in_len']
    lp = torch.log_softmax(torch.from_numpy(logits), -1).numpy()
    B = logits.shape[0]
    tl = (txt != 0).sum(-1)
    nll = np.zeros(B)
    for b in range(B):
        n, g = O.ctc_nll_restated(lp[b, :in_len[b]], txt[b, :tl[b]])
        nll[b] = n
        assert abs(n - z['nll'][b]) < 1e-4
        # torch's gradient of the mean-reduced loss: folded grad / (target_len * B), zero beyond in_len
        np.testing.assert_allclose(g / (tl[b] * B), z['grad'][b, :in_len[b]], atol=2e-6)
        assert np.all(z['grad'][b, in_len[b]:] == 0)
    assert abs(np.mean(nll / tl) - float(z['loss'])) < 1e-5
    # brute force over all alignments for a tiny case
    T, V = 4, 3
    rng = np.random.Generator(np.random.PCG64(3))
    lpt = torch.log_softmax(torch.from_numpy(rng.standard_normal((T, V)).astype(np.float32)), -1).numpy()
    lab = [1, 2]
    tot = -np.inf
    import itertools
    for path in itertools.product(range(V), repeat=T):
        col, prev = [], None
        for s in path:
            if s != prev and s != 0:
                col.append(s)
            prev = s
        if col == lab:
            tot = np.logaddexp(tot, sum(lpt[t, s] for t, s in enumerate(path)))
    n, _ = O.ctc_nll_restated(lpt, lab)
    assert abs(n + tot) < 1e-6
    # infeasible alignment -> inf loss (and the reference reports nan gradients)
    assert np.isinf(float(z['inf_loss'])) and bool(z['inf_grad_isnan'])
    n_inf, g_inf = O.ctc_nll_restated(lp[0, :3], [2, 2, 1])
    assert np.isinf(n_inf) and np.isnan(g_inf).any()


def test_lstm_loop_equals_aten():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 9, 5, generator=g)
    P = {k: torch.randn(s, generator=g) * 0.3 for k, s in {
        'weight_ih_l0': (12, 5), 'weight_hh_l0': (12, 3), 'bias_ih_l0': (12,), 'bias_hh_l0': (12,),
        'weight_ih_l0_reverse': (12, 5), 'weight_hh_l0_reverse': (12, 3), 'bias_ih_l0_reverse': (12,),
        'bias_hh_l0_reverse': (12,)}.items()}
    np.testing.assert_allclose(O.bilstm(x, P, '').numpy(), O.bilstm_aten(x, P, '').numpy(), atol=1e-6)


# ---- inference leg: oracle/decode_oracle.py against the reference's hypotheses (src/ctc.py, src/decode.py, src/lm.py) ----
def _decode_setup(meta, wseed_key='wseed'):
    from oracle import decode_oracle as D
    cfg = O.ModelCfg(meta['model'], meta['D'], meta['V'])
    P = O.seeded_state_dict(O.param_shapes(cfg), meta[wseed_key])
    lm_cfg = meta['lm']
    shapes = {'emb.weight': (meta['V'], lm_cfg['emb_dim'])}
    for l in range(lm_cfg['n_layers']):
        din = lm_cfg['emb_dim'] if l == 0 else lm_cfg['dim']
        shapes['rnn.weight_ih_l%d' % l] = (4 * lm_cfg['dim'], din)
        shapes['rnn.weight_hh_l%d' % l] = (4 * lm_cfg['dim'], lm_cfg['dim'])
        shapes['rnn.bias_ih_l%d' % l] = (4 * lm_cfg['dim'],)
        shapes['rnn.bias_hh_l%d' % l] = (4 * lm_cfg['dim'],)
    if not lm_cfg['emb_tying']:
        shapes['trans.weight'] = (meta['V'], lm_cfg['emb_dim'])
        shapes['trans.bias'] = (meta['V'],)
    P_lm = O.seeded_state_dict(shapes, meta['lm_wseed'])
    return D, cfg, P, (P_lm, lm_cfg)


def test_decode_prefix_scorer_matches_reference(golden_dir):
    from oracle import decode_oracle as D
    meta, z = load(golden_dir, 'g7_decode')
    x = z['ctc_logp']
    cand = z['cand'].tolist()
    r0 = D.ctc_prefix_init(x)
    np.testing.assert_array_equal(r0, z['r0'])
    psi1, r1 = D.ctc_prefix_cheap(x, [], r0, cand)
    psi2, r2 = D.ctc_prefix_cheap(x, [7], r1[cand.index(7)], cand)
    psi3, r3 = D.ctc_prefix_cheap(x, [7, 7], r2[cand.index(7)], [7, 1, 3])
    for got, key in ((psi1, 'psi1'), (r1, 'r1'), (psi2, 'psi2'), (r2, 'r2'), (psi3, 'psi3'), (r3, 'r3')):
        np.testing.assert_array_equal(got, z[key], err_msg=key)          # same float32 numpy arithmetic: bit-exact


@pytest.mark.parametrize('tag,ctc_w,lm_w', [('att', 0.0, 0.0), ('ctc', 0.3, 0.0), ('ctc_lm', 0.3, 0.5)])
def test_decode_beam_search_matches_reference(golden_dir, tag, ctc_w, lm_w):
    meta, z = load(golden_dir, 'g7_decode')
    D, cfg, P, lm = _decode_setup(meta)
    hyps = D.beam_search(torch.from_numpy(z['feat']), torch.from_numpy(z['feat_len']), P, cfg, meta['beam'], meta['min_len_ratio'],
                         meta['max_len_ratio'], ctc_weight=ctc_w, lm=lm if lm_w > 0 else None, lm_weight=lm_w)
    assert len(hyps) == int(z['n_' + tag])
    for i, (seq, scores) in enumerate(hyps):
        assert seq == z['%s_seq%d' % (tag, i)].tolist(), (tag, i)
        np.testing.assert_allclose(np.array(scores, np.float32), z['%s_score%d' % (tag, i)], atol=2e-4, err_msg='%s %d' % (tag, i))


def test_decode_config4_size_matches_reference(golden_dir):
    """BASELINE config 4 at its size: 12 M-parameter model, beam 8, CTC 0.3, 4 x 1024 LM 0.3 (first utterance, T = 400)."""
    meta, z = load(golden_dir, 'g7b_decode_config4')
    D, cfg, P, lm = _decode_setup(meta)
    hyps = D.beam_search(torch.from_numpy(z['feat0']), torch.from_numpy(z['feat_len0']), P, cfg, meta['beam'], meta['min_len_ratio'],
                         meta['max_len_ratio'], ctc_weight=meta['ctc_weight'], lm=lm, lm_weight=meta['lm_weight'])
    assert len(hyps) == int(z['n0'])
    for i, (seq, scores) in enumerate(hyps):
        assert seq == z['u0_seq%d' % i].tolist(), i
        np.testing.assert_allclose(np.array(scores, np.float32), z['u0_score%d' % i], atol=5e-4)
