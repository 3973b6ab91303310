"""One training step on the HIP path, shared by the Solver (bin/train_asr.py) and bench.py:
forward -> CTC + attention losses -> backward -> [gradient all-reduce] -> global-norm clip + NaN guard
+ Adadelta, the sequence bin/train_asr.py:200-253 + src/solver.py:88-106 of the reference runs."""
import torch


def train_step(model, optimizer, ctc_crit, att_crit, feat, feat_len, txt, decode_step, tf_rate=1.0, dp=None,
               clip=5.0, txt_len=None, optimize=True):
    """Returns dict(total_loss, ctc_loss, att_loss, grad_normsq (device float64 scalar)).  No host sync."""
    from src import hipabi as H
    cur = torch.cuda.current_stream()
    if not (feat.is_cuda and H.overlap_enabled() and (dp is None or H.overlap_dp_enabled())) or cur != torch.cuda.default_stream():
        return _train_step(model, optimizer, ctc_crit, att_crit, feat, feat_len, txt, decode_step, tf_rate, dp, clip, txt_len, optimize)
    work = H.work_stream()             # see its docstring: the CU-masked streams serialise against the default stream
    work.wait_stream(cur)
    with torch.cuda.stream(work):
        out = _train_step(model, optimizer, ctc_crit, att_crit, feat, feat_len, txt, decode_step, tf_rate, dp, clip, txt_len, optimize)
    cur.wait_stream(work)
    for v in out.values():
        if torch.is_tensor(v):
            v.record_stream(cur)
    return out


def scale_weight(w_dev, k):
    """k * w_dev for a one-element device tensor (asr_loss_mix with a host constant as the second factor)."""
    from src import hipabi as H
    from src.functions import loss_weight
    if not torch.is_tensor(w_dev):
        return loss_weight(float(w_dev) * k, torch.device('cuda', torch.cuda.current_device()))
    w = w_dev.reshape(1).to(torch.float32).contiguous()
    out = torch.empty(1, dtype=torch.float32, device=w.device)
    H.call('asr_loss_mix', H.ptr(w), H.ptr(loss_weight(k, w.device)), None, None, H.ptr(out), H.stream_ptr())
    return out


def _train_step(model, optimizer, ctc_crit, att_crit, feat, feat_len, txt, decode_step, tf_rate=1.0, dp=None,
                clip=5.0, txt_len=None, optimize=True):
    from src import hipabi as H
    opt = optimizer.opt if hasattr(optimizer, 'opt') else optimizer
    opt.zero_grad()
    if txt_len is None:
        txt_len = torch.sum(txt != 0, dim=-1)
    ctc_output, encode_len, att_output, att_align, _ = model(feat, feat_len, decode_step, tf_rate=tf_rate, teacher=txt, ctc_async=True)
    total, ctc_loss, att_loss = None, None, None
    from src.functions import LossMixFn, loss_weight
    dev = feat.device
    if ctc_output is not None:
        if getattr(ctc_output, '_asr_side', False):       # the CTC branch lives on the side stream (ASR.forward): loss there too
            with H.side_branch(False, txt, txt_len):
                ctc_loss = ctc_crit(ctc_output.transpose(0, 1), txt, encode_len, txt_len)
            H.join_branch(ctc_loss, ctc_output)
        else:
            ctc_loss = ctc_crit(ctc_output.transpose(0, 1), txt, encode_len, txt_len)
    w_att = None
    if att_output is not None:
        b, t, _ = att_output.shape
        att_loss = att_crit(att_output.view(b * t, -1), txt[:, :t].reshape(-1))
        if dp is not None and dp.world > 1:
            # exact global token-mean under data parallelism: the factor is a device scalar out of an all-reduce
            w_att = scale_weight(dp.ce_weight(txt_len.sum()), 1 - model.ctc_weight)
        else:
            w_att = loss_weight(1 - model.ctc_weight, dev)
    # total = w ctc + (1 - w) att (bin/train_asr.py:238,246): one one-thread kernel forward, one per branch backward
    if ctc_loss is not None and att_loss is not None:
        total = LossMixFn.apply(ctc_loss, loss_weight(model.ctc_weight, dev), att_loss, w_att)
    elif ctc_loss is not None:
        total = LossMixFn.apply(ctc_loss, loss_weight(model.ctc_weight, dev), None, None)
    else:
        total = LossMixFn.apply(att_loss, w_att, None, None)
    total.backward()
    H.join_side()          # parameter-gradient work issued on the side stream (no-op when the engine callback already ran)
    grad_mul = 1.0
    if dp is not None:
        dp.finish()
        grad_mul = dp.grad_mul
    normsq = opt.grad_norm()
    if optimize:
        opt.step(clip=clip, grad_mul=grad_mul, use_norm=True)
    return {'total_loss': total.detach(), 'ctc_loss': ctc_loss, 'att_loss': att_loss, 'grad_normsq': normsq,
            'grad_mul': grad_mul, 'ctc_output': ctc_output, 'att_output': att_output, 'att_align': att_align,
            'encode_len': encode_len}
