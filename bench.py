#!/usr/bin/env python
"""Throughput of the joint CTC-attention training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one synthetic LibriSpeech-shaped batch per GPU, inputs already
resident in HBM: forward (4 x BiLSTM-320 encoder, CTC head, location-aware attention decoder) + CTC and
cross-entropy losses + backward + [RCCL all-reduce of the flat gradient] + global-norm clip + Adadelta.
Workload = BASELINE.json configs[1]: config/librispeech_asr.yaml, B=16 x T=1200 frames of 160-dim
fbank+delta per GPU, L=180 tokens, bf16 MFMA contractions with fp32 accumulate, dropout on.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'e2e-asr-pytorch_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import yaml  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='utterances per GPU')
    ap.add_argument('--frames', type=int, default=1200)
    ap.add_argument('--tokens', type=int, default=180)
    ap.add_argument('--prec', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--config', default=os.path.join(PKG, 'config', 'librispeech_asr.yaml'))
    ap.add_argument('--vgg', type=int, default=None, help='override model.encoder.vgg (1: VGGExtractor, 5: VGGExtractor_LN) for the SURVEY D3 variants')
    ap.add_argument('--waveform', action='store_true', help='resident input = 16 kHz waveforms; the GPU fbank (asr_fbank) runs inside the step')
    ap.add_argument('--host-input', action='store_true', help='the batch (fbank, lengths, tokens) is handed over in pinned host memory and copied to the GPU inside every step: the PCIe-inclusive rate of DESIGN.md section 6 (never the headline value)')
    ap.add_argument('--shape', default='fixed', choices=['fixed', 'librispeech'], help="librispeech: 8 buckets per GPU with SURVEY 8d's length model clip(N(1270,480),150,2450) and the reference's halving rule (B = 8 once the longest utterance exceeds 800 frames), cycled through; --frames / --tokens are ignored")
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='process-group backend for --gpus N > 1 (nccl = RCCL over xGMI; gloo only with --dry-run)')
    ap.add_argument('--dry-run', action='store_true', help='start the ranks, form the process group, exchange one all-gather and print the JSON line without touching a GPU (CPU test of the launch path)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    return ap.parse_args()


class KernelTimer(object):
    """HIP-event timing of selected C-ABI calls on the stream they are launched on (torch's current stream)."""

    def __init__(self, names):
        self.names, self.records, self.enabled = set(names), [], False

    def wrap(self, H):
        orig = H.call
        timer = self

        def call(name, *args):
            if timer.enabled and name in timer.names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig(name, *args)
                e1.record()
                timer.records.append((name, args, e0, e1))
            else:
                orig(name, *args)
        H.call = call

    def summary(self):
        out = {}
        for name, args, e0, e1 in self.records:
            out.setdefault(name, []).append((args, e0.elapsed_time(e1)))
        return out


def cpu_baseline(cfg_model, D, V, seconds, B=16, T=1200, L=180):
    """The CPU restatement (oracle, torch fused LSTM on the host cores) on a bounded sample of the same workload: the SAME
    batch shape as the timed GPU step (B x T frames, L tokens; one pass of B=16 x T=1200 takes a few seconds), fwd+bwd,
    repeated for ~`seconds` (at least one full pass)."""
    from oracle import asr_oracle as O
    from src.synthetic import librispeech_shaped_batch
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))     # the GPU box grants a 16-CPU share per GPU; more threads only oversubscribe
    torch.set_num_threads(cores)
    cfg = O.ModelCfg(cfg_model, D, V)
    P = {k: v.requires_grad_(True) for k, v in O.seeded_state_dict(O.param_shapes(cfg), 1).items()}
    feat, lens, txt = librispeech_shaped_batch(B, T, D, L, V, seed=7)
    n, t0 = 0, time.time()
    log('cpu baseline: %d threads' % cores)
    while True:
        res = O.asr_losses(feat, lens, txt, P, cfg, lstm_impl=O.bilstm_aten)
        res['total_loss'].backward()
        for p in P.values():
            p.grad = None
        n += 1
        log('cpu baseline pass %d at %.1f s' % (n, time.time() - t0))
        if time.time() - t0 > seconds or n >= 50:
            break
    dt = time.time() - t0
    return {'value': B * T * n / dt, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': 'oracle (plain PyTorch CPU restatement, fp32, fused ATen LSTM) fwd+bwd on B=%d x T=%d x L=%d, %d passes in %.1f s'
                      % (B, T, L, n, dt)}


def workload_string(config, args, Dfeat):
    if args.shape == 'librispeech':
        return ('config/librispeech_asr.yaml (vgg %d, 4xBiLSTM-320, joint CTC-att 0.5), LibriSpeech-shaped buckets per GPU: %d utterances drawn as '
                'T ~ clip(N(1270,480),150,2450) frames x D=%d, L = 0.14 T tokens, halving rule (B = 8 when the longest > 800 frames), 8 buckets cycled, '
                '%sdelta+SpecAugment on GPU, fwd+losses+bwd+grad-allreduce+clip+Adadelta, dropout on'
                % (config['model']['encoder']['vgg'], args.batch, Dfeat, 'waveform in: GPU fbank + ' if args.waveform else ''))
    return ('config/librispeech_asr.yaml (vgg %d, 4xBiLSTM-320, joint CTC-att 0.5), B=%d x T=%d x D=%d per GPU, L=%d, '
            '%sdelta+SpecAugment on GPU, fwd+losses+bwd+grad-allreduce+clip+Adadelta, dropout on'
            % (config['model']['encoder']['vgg'], args.batch, args.frames, Dfeat, args.tokens, 'waveform in: GPU fbank + ' if args.waveform else ''))


# the PMC files of round 2 carry no workload field: they were collected on the default command
LEGACY_PMC_WORKLOAD = ('config/librispeech_asr.yaml (vgg 0, 4xBiLSTM-320, joint CTC-att 0.5), B=16 x T=1200 x D=160 per GPU, L=180, '
                       'delta+SpecAugment on GPU, fwd+losses+bwd+grad-allreduce+clip+Adadelta, dropout on')


def log(msg):
    print('[bench] ' + msg, file=sys.stderr, flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this script under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and exit with their status.  Nothing has touched a GPU in this process, so the
    children are ordinary child processes, not a re-exec of a process that owns the device."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('no WORLD_SIZE in the environment: starting %d ranks: %s' % (args.gpus, ' '.join(cmd)))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """Launch-path check without a GPU: every rank reports in through one all-gather; rank 0 prints the line."""
    import torch.distributed as dist
    ranks = [rank]
    if world > 1:
        mine = torch.tensor([rank], dtype=torch.int64)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        dist.barrier()
        ranks = sorted(int(x) for x in got)
    if rank == 0:
        print(json.dumps({'metric': 'audio frames/sec (fwd+bwd) LibriSpeech-100 joint CTC-att', 'value': None, 'dry_run': True,
                          'n_gpus': world, 'ranks': ranks, 'backend': args.backend if world > 1 else None,
                          'steps': args.steps, 'warmup': args.warmup}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.backend == 'gloo' and not args.dry_run:
        sys.exit('--backend gloo is for --dry-run only: the step runs on the GPU over RCCL')
    from src import dist as D_
    rank, world, local = D_.init_from_env(args.backend if args.gpus > 1 else None)
    if world != args.gpus:
        sys.exit('bench.py --gpus %d was started with WORLD_SIZE=%d: the process group must have exactly one rank per GPU asked for' % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    torch.cuda.set_device(local)
    from src import hipabi as H
    from src.asr import ASR
    from src.optim import Optimizer
    from src.step import train_step
    from src.synthetic import librispeech_shaped_batch
    from src.util import CTCLoss, CrossEntropyLoss, LabelSmoothingLoss

    config = yaml.safe_load(open(args.config))
    if args.vgg is not None:
        config['model']['encoder']['vgg'] = args.vgg
    Dfeat = config['data']['audio']['feat_dim'] * (config['data']['audio']['delta_order'] + 1)
    V = 31
    torch.manual_seed(0)
    model = ASR(Dfeat, V, args.batch, prec=args.prec, seed=1234 + rank, **config['model']).cuda().train()
    dp = None
    if world > 1:
        dp = model.attach_data_parallel()
        dp.broadcast_params(0)
    hp = config['hparas']
    optimizer = Optimizer(model.parameters(), hp['optimizer'], hp['lr'], hp['eps'], hp.get('lr_scheduler'),
                          hp.get('tf_start', 1), hp.get('tf_end', 1), hp.get('tf_step', 1))
    ctc_crit = CTCLoss(blank=0, zero_infinity=False)
    att_crit = LabelSmoothingLoss(31, 0.1) if hp.get('label_smoothing', False) else CrossEntropyLoss(ignore_index=0)
    B, T, L = args.batch, args.frames, args.tokens
    nmel = config['data']['audio']['feat_dim']
    if args.shape == 'librispeech':
        from src.synthetic import librispeech_length_batches
        assert not args.waveform and not args.host_input, '--shape librispeech takes resident fbank batches'
        batches = [(f, fl, tx, (tx != 0).sum(-1), tx.shape[1])
                   for f, fl, tx in librispeech_length_batches(8, nmel, V, seed=1234 + rank, batch_size=B, device='cuda')]
    else:
        f_, fl_, tx_ = librispeech_shaped_batch(B, T, nmel, L, V, seed=1234 + rank, device='cuda')
        batches = [(f_, fl_, tx_, (tx_ != 0).sum(-1), L)]
    fbank, feat_len, txt, txt_len, _ = batches[0]
    step_no = [0]
    # the 80-dim fbank batch is the resident input; delta stacking + SpecAugment (data.audio.augment) run on the GPU
    from src.audio import Delta, Augment, ExtractAudioFeature
    wav = wav_len = fb_mod = None
    if args.waveform:
        # waveforms whose frame counts are the batch's lengths: N = (T-1)*hop + 1 samples (src/data.SyntheticLoader)
        a = config['data']['audio']
        fb_mod = ExtractAudioFeature(mode=a['feat_type'], num_mel_bins=nmel, frame_length=a.get('frame_length', 25),
                                     frame_shift=a.get('frame_shift', 10), ref_level_db=a.get('ref_level_db', 20),
                                     min_level_db=a.get('min_level_db', -100), preemphasis_coeff=a.get('preemphasis_coeff', 0.97)).cuda()
        wav_len = (feat_len - 1) * fb_mod.hop + 1
        g = torch.Generator(device='cuda').manual_seed(99 + rank)
        wav = 0.1 * torch.randn(B, int(wav_len.max()), device='cuda', generator=g)
        wav = wav * (torch.arange(wav.shape[1], device='cuda')[None, :] < wav_len[:, None])
    delta = Delta(config['data']['audio']['delta_order'], config['data']['audio'].get('delta_window_size', 2)).cuda() \
        if config['data']['audio']['delta_order'] >= 1 else None
    augment = Augment(seed=1234 + rank).cuda() if config['data']['audio'].get('augment', False) else None

    # inside the timed region only the dominant kernel (the LSTM recurrence, 8 launches per step) carries HIP events; the
    # other stages and the contractions are timed in a short pass AFTER it (an event pair per call costs host time and
    # ~4 us of queue time: 100 pairs per step made the step 0.3-1.5 ms longer, depending on the host)
    timer = KernelTimer(['asr_lstm_fwd', 'asr_lstm_bwd', 'asr_lstm16_fwd', 'asr_lstm16_bwd'] + (['asr_fbank'] if args.waveform else []))
    timer.wrap(H)

    import contextlib

    use_work = H.overlap_enabled() and (world == 1 or H.overlap_dp_enabled())
    if use_work:
        # model init, flatten(), broadcast_params and the optimizer state were issued on the default stream; the work stream
        # is non-blocking and does not serialise with it
        H.work_stream().wait_stream(torch.cuda.current_stream())

    def step():
        # with the CU-masked overlap the whole step (input transform included) runs on the non-default work stream
        ctx = torch.cuda.stream(H.work_stream()) if use_work else contextlib.nullcontext()
        with ctx:
            return step_()

    host_batch = None
    if args.host_input:
        host_batch = tuple(t.cpu().pin_memory() for t in (fbank, feat_len, txt, txt_len))

    def step_():
        nonlocal fbank, feat_len, txt, txt_len
        fbank, feat_len, txt, txt_len, L = batches[step_no[0] % len(batches)]
        step_no[0] += 1
        if host_batch is not None:
            fbank, feat_len, txt, txt_len = (t.to('cuda', non_blocking=True) for t in host_batch)
        feat = fbank
        if fb_mod is not None:
            feat, _ = fb_mod(wav, wav_len)
        if delta is not None:
            feat, _ = delta(feat, feat_len)
        elif augment is not None:
            feat = feat.clone()
        if augment is not None:
            feat, _ = augment(feat, feat_len)
        return train_step(model, optimizer, ctc_crit, att_crit, feat, feat_len, txt, L, tf_rate=1.0, dp=dp, clip=5.0,
                          txt_len=txt_len)

    log('model built (%d params), warm-up...' % sum(p.numel() for p in model.parameters()))
    for i in range(max(args.warmup, len(batches) if len(batches) > 1 else 0)):      # every bucket shape once: plans, work areas
        out = step()
        torch.cuda.synchronize()
        log('warm-up step %d done, loss %.4f' % (i, float(out['total_loss'])))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    step_no[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    t_host = time.perf_counter() - t0           # host time to ENQUEUE the steps (diagnostic: how far the CPU runs ahead of the GPU)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = False
    per_rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        # every rank's own wall time per step (load balance across ranks explains the first scaling run) and the maximum
        mine = torch.tensor([dt], dtype=torch.float64, device='cuda')
        allt = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allt, mine)
        per_rank_ms = [float(x) / args.steps * 1e3 for x in allt]
        dt = max(float(x) for x in allt)
    loss = float(out['total_loss'])
    H.raise_if_aborted()          # a persistent launch that gave up inside the timed region invalidates the run
    log('timed %d steps in %.3f s' % (args.steps, dt))
    lstm_summ = timer.summary()
    post_steps = 3
    timer.names, timer.records, timer.enabled = {'asr_att_decoder_fwd', 'asr_att_decoder_bwd', 'asr_att_decoder_bwd_ex', 'asr_att_decoder_bwd_params', 'asr_gemm', 'asr_gemm16',
                                                    'asr_conv3x3', 'asr_conv3x3_16', 'asr_conv3x3_16_wgrad'}, [], True
    for _ in range(post_steps):
        step()
    torch.cuda.synchronize()
    timer.enabled = False
    post_summ = timer.summary()
    # padded and valid input frames of the timed steps on THIS rank (buckets cycled in order), times the ranks
    seen = [batches[i % len(batches)] for i in range(args.steps)]
    frames = sum(int(bt[0].shape[0] * bt[0].shape[1]) for bt in seen) * world
    valid = sum(int(bt[1].sum()) for bt in seen) * world
    if rank != 0:
        return

    # ---- roofline of the dominant kernel: the persistent encoder-LSTM recurrence (one launch per layer and pass) ---
    summ = lstm_summ
    enc = config['model']['encoder']
    Hd, ND = enc['dim'][0], 2 if enc['bidirection'] else 1
    tot = {k: sum(ms for _, ms in v) for k, v in summ.items()}
    roof = None
    fast = 'asr_lstm16_bwd' in summ or 'asr_lstm16_fwd' in summ          # bf16-storage recurrence (lstm_persist3.hip)
    fn_f, fn_b = ('asr_lstm16_fwd', 'asr_lstm16_bwd') if fast else ('asr_lstm_fwd', 'asr_lstm_bwd')
    bwd_dom = tot.get(fn_b, 0) >= tot.get(fn_f, 0)
    calls = summ.get(fn_b if bwd_dom else fn_f, [])
    gen = 'p3' if fast else 'p2'
    name = ('lstm_bwd_' if bwd_dom else 'lstm_fwd_') + gen
    if calls:
        # algorithmic bytes of ONE launch (all T steps of one layer, both directions), per (b,t):
        #   fp32 storage (p2): forward  gate pre-activations read + activated gates written (2*ND*4H), h and c written (2*ND*H), x4 B
        #                      backward dy, gates, c, c_prev read (ND*H + ND*4H + 2*ND*H), gate gradients written (ND*4H), x4 B
        #   bf16 storage (p3): gates / h / dy are 2 B per element, the cell state stays fp32
        # plus W_hh (fp32 master) once
        if fast:
            tidx = 5
            per_bt = (2 * ND * Hd + 2 * 2 * ND * 4 * Hd + 4 * 2 * ND * Hd) if bwd_dom else (2 * 2 * ND * 4 * Hd + 2 * ND * Hd + 4 * ND * Hd)
        else:
            tidx = 5 if bwd_dom else 6
            per_bt = 4 * ((ND * Hd + 2 * ND * 4 * Hd + 2 * ND * Hd) if bwd_dom else (2 * ND * 4 * Hd + 2 * ND * Hd))
        nbytes = sum(1.0 * (a[tidx - 1] * a[tidx] * per_bt + 4 * ND * 4 * Hd * Hd) for a, _ in calls)     # a[tidx - 1] = B, a[tidx] = T of the launch
        secs = sum(ms for _, ms in calls) * 1e-3
        steps_total = sum(a[tidx] for a, _ in calls)
        # PMC traffic is quoted only when the committed collection was made on THIS workload (tools/collect_profiles.sh stores
        # the bench line's config.workload in the file); on any other workload it is null, never another shape's number
        traffic, traffic_src = None, None
        for tname in ('r03_pmc_traffic.json', 'r02_pmc_traffic.json'):
            tpath = os.path.join(ROOT, 'profiles', tname)
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                same = tj.get('_workload', LEGACY_PMC_WORKLOAD) == workload_string(config, args, Dfeat)
                traffic = tj.get(name, {}).get('hbm_bytes_per_launch') if same else None
                if traffic is not None:
                    traffic_src = 'profiles/' + tname + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_traffic.py)'
                    break
        roof = {'kernel': name, 'bound': 'hbm', 'achieved': nbytes / secs / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                'frac': nbytes / secs / 1e9 / 8000.0, 'traffic': traffic, 'traffic_source': traffic_src,
                'avg_launch_ms': secs * 1e3 / len(calls), 'launches_per_step': len(calls) / args.steps,
                'algorithmic_bytes_per_launch': nbytes / len(calls), 'us_per_time_step': secs * 1e6 / steps_total,
                'us_per_time_step_fwd': (sum(ms for _, ms in summ.get(fn_f, [])) * 1e3 / max(1, sum(a[5 if fast else 6] for a, _ in summ.get(fn_f, [])))),
                'us_per_time_step_bwd': (sum(ms for _, ms in summ.get(fn_b, [])) * 1e3 / max(1, sum(a[5] for a, _ in summ.get(fn_b, [])))),
                'note': 'latency-bound recurrence: the figure of merit is us_per_time_step (inter-workgroup hand-off), not GB/s'}
    # ---- the MFMA-bound part: every contraction the host issues (encoder input projections and projections, CTC / key /
    #      vocabulary heads; forward, input gradients, split-K weight gradients): 2*M*N*K*batch flop per call over its
    #      HIP-event time, against the dense bf16 peak (fp32 mode: the same kernel on the fp32 MFMA path)
    gemm = None
    gcalls = post_summ.get('asr_gemm', []) + post_summ.get('asr_gemm16', [])
    n16 = len(post_summ.get('asr_gemm16', []))
    if gcalls:
        flop = sum(2.0 * a[4] * a[5] * a[6] * max(1, a[15]) for a, _ in post_summ.get('asr_gemm', []))
        flop += sum(2.0 * a[4] * a[5] * a[6] for a, _ in post_summ.get('asr_gemm16', []))
        gsec = sum(ms for _, ms in gcalls) * 1e-3
        gemm = {'kernel': 'every contraction the host issues: %d of %d calls per step are gemm16_nt / gemm16_tn (128x128x64 tiles, bf16 operands direct to LDS, v_mfma_f32_16x16x32_bf16), the rest gemm_kernel (128x128x32, fp32 operands converted while staging)' % (n16 // post_steps, len(gcalls) // post_steps),
                'bound': 'mfma', 'achieved': flop / gsec / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s',
                'frac': flop / gsec / 1e12 / 2500.0, 'calls_per_step': len(gcalls) / post_steps,
                'ms_per_step': gsec * 1e3 / post_steps, 'gflop_per_step': flop / post_steps / 1e9,
                'measured': '%d steps after the timed region' % post_steps,
                }
    # ---- the VGG convolutions (vgg 1 / vgg 5 workloads): implicit GEMMs, 2 * pixels * N * 9C flop per call, against the same peak
    conv = None
    ccalls = post_summ.get('asr_conv3x3', []) + post_summ.get('asr_conv3x3_16', []) + post_summ.get('asr_conv3x3_16_wgrad', [])
    if ccalls:
        cflop = 0.0
        for a, _ in post_summ.get('asr_conv3x3', []):            # (img, w, out, bias, B, T, F, Ci, Co, mode, ...)
            cflop += 2.0 * a[4] * a[5] * a[6] * a[7] * a[8] * 9
        for a, _ in post_summ.get('asr_conv3x3_16', []):         # (img, w, out, bias, B, T, F, C, N, K, implicit, ...): bordered pixel grid
            cflop += 2.0 * a[4] * (a[5] + 2) * (a[6] + 2) * a[8] * a[9]
        for a, _ in post_summ.get('asr_conv3x3_16_wgrad', []):   # (img, dout, dw, B, T, F, C, N, ...)
            cflop += 2.0 * a[3] * (a[4] + 2) * (a[5] + 2) * a[6] * a[7] * 9
        csec = sum(ms for _, ms in ccalls) * 1e-3
        conv = {'kernel': 'VGG 3x3 convolutions: gemm16_nt_kernel<2,2,4,CONV> (implicit GEMM on zero-bordered bf16 images) forward + input gradient, '
                          'gemm16_tn_kernel with the tap as a grid dimension for the weight gradient (fp32 mode / uncovered shapes: gemm_kernel)',
                'bound': 'mfma', 'achieved': cflop / csec / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': cflop / csec / 1e12 / 2500.0,
                'calls_per_step': len(ccalls) / post_steps, 'ms_per_step': csec * 1e3 / post_steps, 'gflop_per_step': cflop / post_steps / 1e9}
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        log('cpu baseline (oracle) for ~%.0f s...' % args.cpu_seconds)
        cpu = cpu_baseline(config['model'], Dfeat, V, args.cpu_seconds, min(B, 16), args.frames, args.tokens)
    line = {
        'metric': 'audio frames/sec (fwd+bwd) LibriSpeech-100 joint CTC-att',
        'value': frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.prec, 'data': 'synthetic',
        'config': {'workload': workload_string(config, args, Dfeat),
                   'global_batch': (B if args.shape == 'fixed' else [int(bt[0].shape[0]) for bt in batches]) if world == 1 else B * world,
                   'frames_per_utt': T if args.shape == 'fixed' else [int(bt[0].shape[1]) for bt in batches], 'parallelism': 'dp%d' % world},
        'valid_frames_per_s': valid / dt, 'loss': loss, 'per_rank_ms_per_step': per_rank_ms,
        'per_rank_frames': [frames // world] * world, 'per_rank_max_T': [max(int(bt[0].shape[1]) for bt in batches)] * world, 'host_enqueue_ms_per_step': t_host / args.steps * 1e3,
        'stage_ms_per_step': dict([(k, v / args.steps) for k, v in tot.items()] +
                                  [(k, sum(ms for _, ms in v) / post_steps) for k, v in post_summ.items() if k not in ('asr_gemm', 'asr_gemm16', 'asr_conv3x3', 'asr_conv3x3_16', 'asr_conv3x3_16_wgrad')]),
        'roofline': roof, 'roofline_gemm': gemm, 'roofline_conv': conv, 'cpu_baseline': cpu,
    }
    if args.host_input:
        line['input'] = 'pinned host memory, copied to the GPU inside every step (PCIe-inclusive; not the headline value)'
    print(json.dumps(line))


if __name__ == '__main__':
    main()
