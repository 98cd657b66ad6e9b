#!/usr/bin/env python
"""Headline benchmark: training clips/sec of ViT-B/16 + AVT-h (10 x 224^2 frames, C = 3806) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the caller launches the ranks (``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment) or
``python bench.py --gpus N`` spawns them itself (re-exec under torch.distributed.run on 127.0.0.1, the counterpart of the
reference's one-process-per-GPU launch, train_net.py:43 / common/utils.py:106-150); it exits non-zero with a message when
the node has fewer than N devices.

A step = forward + losses + backward (+ overlapped RCCL gradient all-reduce for N > 1) + fused SGD-nesterov update on a
synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line:
  value      whole-job clips/sec (N x per-GPU batch / max-over-ranks step time)
  roofline   bound = mfma.  ``frac`` is the WHOLE-STEP fraction north_star names: clips/s/GPU x algorithmic GFLOP/clip
              / 2.5 PFLOP/s dense bf16.  ``dominant_kernel`` is the bf16 MFMA GEMM family measured live: 2*M*N*K
             of every launch in the timed region / their summed HIP-event durations.  ``traffic`` = HBM bytes per step from
             the committed rocprofv3 PMC pass over the same workload (profiles/pmc_step.json; null when none matches)
  also       (N = 1) short runs of BASELINE configs 4 and 5 at their own architecture on this GPU, carried inside the same line:
             T = 15 frames (128 clips) and ViT-L/16 (96 clips), config 2 at 64 clips and at the reference's own 3 clips per GPU,
             2 warm-up + 5 timed steps each -> value, ms_per_step, frac
  comm       (N > 1) per-rank exchange accounting, the RCCL / NCCL environment knobs in effect, and the GEMM family's time per
             step with and without collectives in flight (CU contention from RCCL's kernels shows up as the difference)
  cpu_baseline  the fp32 CPU oracle (a port of the reference's timm/HF path, oracle/avt_oracle.py) timed on this box's
             host cores on a bounded sample (B = 1 clip, 1 warm-up + 3 timed steps of fwd+bwd+SGD; min/max reported)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0          # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
NUM_CLASSES = 3806

VIT = {'vit_base_patch16_224': (768, 12, 12), 'vit_large_patch16_224': (1024, 24, 16)}


def flops_per_clip(D, L, T, Dh=2048, Lh=6, C=NUM_CLASSES, S=197):
    """GEMM + attention flops, backward = 2x forward except the patch-embed data gradient."""
    patch = 2 * 196 * 768 * D
    f_v = patch + L * (2 * S * D * 3 * D + 4 * S * S * D + 2 * S * D * D + 4 * S * D * 4 * D)
    f_h = 2 * T * D * Dh * 2 + Lh * (2 * T * Dh * 3 * Dh + 4 * T * T * Dh + 2 * T * Dh * Dh + 16 * T * Dh * Dh) + 2 * (T + 1) * D * C
    return 3 * (T * f_v + f_h) - T * patch


def flops_skipped_per_clip(D, T, S=197):
    """Work of the LAST ViT block that the CLS-only evaluation never executes (timm consumes x[:, 0] only):
    the full block minus {k|v projection on all tokens, q / proj / MLP on one token, single-query attention}; x3 for
    forward + backward.  ``flops_per_clip`` (SURVEY 8d's numerator) still counts it; ``executed_frac`` does not."""
    full = 2 * S * D * 3 * D + 4 * S * S * D + 2 * S * D * D + 4 * S * D * 4 * D
    done = 2 * S * D * 2 * D + 2 * D * D + 4 * S * D + 2 * D * D + 4 * D * 4 * D
    return 3 * T * (full - done)


def algorithmic_bytes_per_step(D, L, T, B, Dh=2048, Lh=6, S=197, folded=None):
    """HBM bytes per step per GPU of the implemented dataflow if every tensor crossed HBM exactly as often as the kernel
    sequence consumes / produces it (DESIGN.md section 4 lists the per-kernel terms).  Unit u = one [tokens, D] bf16 tensor.
    A full ViT block moves 26 u forward (qkv 4, attention 4, proj 3, fc1 9, fc2 6 -- since round 5 the two LayerNorms are folded into qkv /
    fc1 and write nothing: 30 u before) and 52 u backward (fc2 w/dgrad 14, fc1 w/dgrad 10, LN2 4, proj w/dgrad 4, attention 8, qkv w/dgrad 8,
    LN1 4); the CLS-only last block 20 u;
    patch embedding: fp32 frames once + 6 u; parameters: 38 B each (bf16 shadow read by forward and dgrad, fp32 gradient
    read-modify-write, 26 B in the fused SGD); the temporal head's activations with u_h = [B*T, Dh] bf16.
    ``folded``: is the LayerNorm fold taken at this size (HipViT.fold_min_rows)?  None = ask the product's rule.  Without it a full block moves 30 u
    forward (a normalised copy written and read per LayerNorm): 82 u in all (round-5 advisor: the 3-clip `also` run was priced with the folded 78 u)."""
    u = B * T * S * D * 2
    if folded is None:
        from avt_amd.models.vit import HipViT, use_fold
        folded = use_fold(HipViT, B * T * S, D, L - 1)
    vit = ((L - 1) * (78 if folded else 82) + 20 + 6) * u + B * T * 3 * 224 * 224 * 4
    n_vit = 768 * D + D + S * D + D + L * (12 * D * D + 13 * D) + 2 * D
    n_head = 2 * D * Dh + 1024 * Dh + Lh * (12 * Dh * Dh + 13 * Dh) + 2 * Dh
    n_cls = (D + 1) * NUM_CLASSES
    head = Lh * 82 * (B * T * Dh * 2)
    return vit + head + 38 * (n_vit + n_head + n_cls)


def build(args, device, world):
    from avt_amd.config import Cfg
    from avt_amd.func.train import Trainer, synthetic_batch
    from avt_amd.func.train_eval_ops import Basic
    from avt_amd.models.base_model import BaseModel
    from avt_amd.optim import FusedSGD
    fp = Cfg(_target_='models.future_prediction.AVTh', n_head=4, n_layer=6, output_len=1, inter_dim=2048,
             return_past_too=True, avg_last_n=1, future_pred_loss=Cfg(_target_='torch.nn.MSELoss'), future_pred_loss_wt=1.0)
    mcfg = Cfg(backbone=Cfg(_target_='models.video_classification.TIMMModel', model_type=args.model),
               backbone_last_n_modules_to_drop=0, backbone_dim=VIT[args.model][0], intermediate_featdim=None,
               temporal_aggregator=Cfg(_target_='models.temporal_aggregation.Identity'),
               temporal_aggregator_after_future_pred=Cfg(_target_='models.temporal_aggregation.Identity'),
               future_predictor=fp, classifier=Cfg(_target_='torch.nn.Linear', bias=True), same_temp_agg_dim=False,
               project_dim_for_nce=None, dropout=0.2, use_cls_mappings=False, classifier_on_past=True,
               add_regression_head=False, bn=Cfg(eps=0.001, mom=0.1))
    if os.environ.get('AVT_FOLD_LN') is not None:          # lab A/B: the LayerNorm kernels in front of qkv / fc1 instead of the fold
        from avt_amd.models.vit import HipViT
        HipViT.fold_layernorm = os.environ['AVT_FOLD_LN'] != '0'
    torch.manual_seed(42)
    model = BaseModel(mcfg, {'action': NUM_CLASSES}, {}).to(device)
    with torch.no_grad():                       # ViT weights: N(0, 0.02) stand-in for the (absent) pretrained checkpoint
        for n, p in model.backbone.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.02)
    opt = FusedSGD(model.parameters(), lr=1e-4 * world, momentum=0.9, nesterov=True, weight_decay=1e-6, arena=model.arena)
    op = Basic(model, device, None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
    trainer = Trainer(model, op, opt, None, {'cls_action': 1.0, 'past_cls_action': 1.0, 'feat': 1.0}, distributed=world > 1,
                      bucket_bytes=args.bucket_mb << 20, reduce_mode=args.reduce_mode, reduce_transport=args.reduce_transport,
                      wire_dtype=torch.bfloat16 if args.wire_dtype == 'bf16' else torch.float32,
                      tail_bytes=None if args.tail_mb < 0 else args.tail_mb << 20)
    rank = int(os.environ.get('RANK', 0))
    data = synthetic_batch(args.batch, args.frames, NUM_CLASSES, device, seed=42 + rank)
    return trainer, data


def cpu_baseline(args):
    """fp32 CPU oracle, fwd + bwd + SGD-nesterov, B = 1 clip (bounded sample of the same workload)."""
    from oracle import avt_oracle as O
    # threads actually used (reported as `cores`).  Not every hardware thread: with all 256 of the GPU box's the fp32 oracle runs
    # 70x SLOWER (0.0051 clips/s, 13 minutes for the four sample steps -- oversubscribed small ops) than with 64 (0.34-0.37 clips/s)
    host_threads = os.cpu_count() or 1
    cores = min(host_threads, 64)
    torch.set_num_threads(cores)
    D, L, H = VIT[args.model]
    orc = O.OracleBaseModel(O.OracleTIMMModel(vit=O.OracleViT(D, L, H)),
                            O.OracleAVTh(D, inter_dim=2048, n_layer=6, n_head=4), D, {'action': NUM_CLASSES}, dropout=0.2)
    orc.train()
    opt = torch.optim.SGD(orc.parameters(), lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-6)
    g = torch.Generator().manual_seed(42)
    B = 1
    video = torch.rand((B, args.frames, 3, 1, 224, 224), generator=g) * 2 - 1
    target = torch.randint(0, NUM_CLASSES, (B,), generator=g)
    sub = torch.randint(-1, NUM_CLASSES, (B, args.frames, 1), generator=g)
    wts = {'cls_action': 1.0, 'past_cls_action': 1.0, 'feat': 1.0}
    times = []
    for it in range(4):
        t0 = time.time()
        out, aux = orc(video, target_shape=target.shape)
        losses, _ = O.basic_loss_accuracy(out, {'action': target}, {'action': sub})
        losses.update(aux)
        tot = O.total_loss(losses, wts)
        opt.zero_grad()
        tot.backward()
        opt.step()
        times.append(time.time() - t0)
    timed = times[1:]
    t = sum(timed) / len(timed)
    return {'value': round(B / t, 4), 'unit': 'clips/s', 'cores': cores, 'host_threads': host_threads, 'kind': 'port',
            'min': round(B / max(timed), 4), 'max': round(B / min(timed), 4),
            'sample': f'B={B} clip x {args.frames} frames, 1 warm-up + {len(timed)} timed fwd+bwd+SGD steps of the fp32 oracle '
                      f'(mean {t:.2f} s/step, range {min(timed):.2f}-{max(timed):.2f} s, '
                      f'{flops_per_clip(D, L, args.frames) * B / t / 1e9:.0f} GFLOP/s)'}


def self_launch(args, argv):
    """``python bench.py --gpus N`` without a launcher: spawn one rank per GPU under torch.distributed.run (127.0.0.1
    rendezvous on a free port) and pass rank 0's JSON line through.  Fails loudly when the node has fewer devices."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and args.backend == 'nccl':
        raise SystemExit(f'bench.py --gpus {args.gpus}: this node exposes {ndev} GPU(s); refusing to oversubscribe '
                         f'(RCCL needs one device per rank)')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def pmc_traffic(args):
    """HBM bytes per step from the committed PMC pass over the same workload (profiles/pmc_step.json), else None."""
    path = os.path.join(ROOT, 'profiles', 'pmc_step.json')
    try:
        with open(path) as f:
            recs = json.load(f)
    except (OSError, ValueError):
        return None
    for r in recs:
        if (r.get('model'), r.get('batch'), r.get('frames')) == (args.model, args.batch, args.frames):
            return r
    return None


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='clips per GPU (weak scaling: fixed per GPU; 256 clips = 2560 frames keep ~150 GB of activations resident)')
    ap.add_argument('--frames', type=int, default=10)
    ap.add_argument('--model', default='vit_base_patch16_224', choices=list(VIT))
    ap.add_argument('--bucket-mb', type=int, default=64)
    ap.add_argument('--reduce-transport', default='torch', choices=['torch', 'abi'], help="who issues the collectives: torch.distributed (backend nccl = RCCL), or the library's own RCCL entry points (avt_allreduce_bucket ..., include/avt_hip.h; needs --backend nccl's one device per rank)")
    ap.add_argument('--reduce-mode', default='all_reduce', choices=['all_reduce', 'rs_ag'], help='gradient exchange per bucket: RCCL all-reduce, or reduce-scatter + all-gather')
    ap.add_argument('--wire-dtype', default='fp32', choices=['fp32', 'bf16'], help='dtype of the gradient buckets on the wire (bf16 halves the xGMI bytes; the sum then keeps 8 mantissa bits)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help="'gloo' lets several ranks share one GPU (a functional check of the N > 1 path on a 1-GPU box; never a measurement)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gemm-trace', action='store_true')
    ap.add_argument('--trace-steps', type=int, default=5, help='steps of the second, untimed-for-the-headline run that records a HIP-event pair around every GEMM launch (roofline.dominant_kernel)')
    ap.add_argument('--graph', action='store_true', help='replay the step from a hipGraph (avt_amd/func/graph.py::CapturedStep: device-resident dropout seeds and learning rates; single process only) -- what a launch-bound small batch wants; the default line stays eager')
    ap.add_argument('--no-also', action='store_true', help='skip the short T = 15 / ViT-L runs that the default N = 1 line carries in "also"')
    ap.add_argument('--tail-mb', type=int, default=-1, help='the last (exposed) exchange is at most this many MiB of gradients (default: half a bucket)')
    ap.add_argument('--nccl-max-nchannels', type=int, default=0, help='export NCCL_MAX_NCHANNELS before RCCL starts: fewer channels = fewer CUs taken from the GEMMs (0 = leave the environment alone)')
    ap.add_argument('--nccl-min-nchannels', type=int, default=0, help='export NCCL_MIN_NCHANNELS before RCCL starts (0 = leave the environment alone)')
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)

    if args.nccl_max_nchannels > 0:
        os.environ['NCCL_MAX_NCHANNELS'] = str(args.nccl_max_nchannels)           # read by RCCL when the communicator is created (below)
    if args.nccl_min_nchannels > 0:
        os.environ['NCCL_MIN_NCHANNELS'] = str(args.nccl_min_nchannels)
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'RANK' not in os.environ:
        raise SystemExit(self_launch(args, argv))
    if args.gpus != world_env:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks')

    import torch.distributed as dist
    from avt_amd import ops
    from avt_amd.common import utils
    dist_on, rank, world, local = utils.init_distributed_mode(args.backend)
    if args.backend == 'gloo':
        local %= max(torch.cuda.device_count(), 1)             # ranks share the devices that exist
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(a, steps, warmup, probe_no_comm=False):
        """Build the model of configuration ``a``, run ``warmup`` untimed and ``steps`` timed steps; returns the raw numbers."""
        trainer, data = build(a, device, world)
        step = trainer.step
        if getattr(a, 'graph', False) and trainer.reducer is None:
            from avt_amd.func.graph import CapturedStep
            step = CapturedStep(trainer, data).step          # (two eager steps on the capturing stream, then the recording)
        for _ in range(warmup):
            step(data)
        sync()
        no_comm = None
        if probe_no_comm and trainer.reducer is not None and not a.no_gemm_trace:
            # the same step with the gradient exchange switched off: the GEMM family's time without RCCL's kernels on the CUs
            # (3 untimed steps; the replicas drift apart meanwhile, so rank 0's parameters are broadcast again afterwards)
            trainer.reducer.paused = True
            ops.GEMM_TRACE = tr = []
            for _ in range(3):
                trainer.step(data)
            sync()
            ops.GEMM_TRACE = None
            trainer.reducer.paused = False
            no_comm = tr
            trainer.reducer.broadcast_parameters(trainer.model)
            trainer.reducer.broadcast_optimizer_state(trainer.optimizer)      # (the momentum drifted apart as well)
            trainer.step(data)
            sync()
        calls0 = _abi.N_CALLS
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _, _, _ = step(data)
        host_enqueue = time.perf_counter() - t0          # the Python side is done enqueuing; the GPU may still be running
        sync()
        elapsed_local = time.perf_counter() - t0
        abi_calls = _abi.N_CALLS - calls0
        # the per-GEMM HIP events (two records per launch) stay out of the timed region: a second, short run of the same step carries them
        trace, trace_elapsed = None, 0.0
        if not a.no_gemm_trace and a.trace_steps > 0:
            ops.GEMM_TRACE = trace = []
            t1 = time.perf_counter()
            for _ in range(a.trace_steps):
                trainer.step(data)
            sync()
            trace_elapsed = time.perf_counter() - t1
            ops.GEMM_TRACE = None
        res = {'elapsed_local': elapsed_local, 'host_enqueue': host_enqueue, 'abi_calls': abi_calls, 'trace': trace,
               'trace_steps': a.trace_steps, 'trace_elapsed': trace_elapsed,
               'no_comm_trace': no_comm, 'loss': float(loss),
               'comm': trainer.reducer.stats(last=steps) if trainer.reducer is not None else None}
        del trainer, data
        import gc
        gc.collect()                              # (autograd nodes of the last step hold the saved activations until collected)
        torch.cuda.empty_cache()
        return res

    from avt_amd import lib as _abi
    m = measure(args, args.steps, args.warmup, probe_no_comm=True)
    elapsed_local, host_enqueue, abi_calls, trace, comm, loss_val = (m['elapsed_local'], m['host_enqueue'], m['abi_calls'], m['trace'],
                                                                      m['comm'], m['loss'])
    comm_all = [comm]
    if dist_on:
        comm_all = [None] * world
        dist.all_gather_object(comm_all, comm)
    # who took part: a device-side sum of ones over the job's communicator (with backend nccl = RCCL: the ranks RCCL actually connected)
    # and each rank's device identity (N ranks on N distinct devices, or -- gloo functional check -- sharing some)
    ranks_seen, devices_seen = 1, None
    if dist_on:
        one = torch.ones(1, device=device, dtype=torch.float32)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        devices_seen = [None] * world
        pr = torch.cuda.get_device_properties(device)
        dist.all_gather_object(devices_seen, f'{pr.name} #{local} pci {getattr(pr, "pci_bus_id", "?")}:{getattr(pr, "pci_device_id", "?")}')
    t = torch.tensor([elapsed_local], device=device, dtype=torch.float64)
    per_rank = [t.clone() for _ in range(world)]
    if dist_on:
        dist.all_gather(per_rank, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)

    def gemm_rows(tr, steps, step_s):
        """Group the traced GEMM launches by (kernel template, M, N, K): [(name, shape, launches/step, ms/step, TFLOP/s)]."""
        groups = {}
        for name, fl, e0, e1, shape in tr:
            d = groups.setdefault((name, shape), [0.0, 0.0, 0])
            d[0] += fl; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1
        return [{'kernel': k[0], 'MNK': list(k[1]), 'launches_per_step': round(v[2] / steps, 1), 'ms_per_step': round(v[1] / steps * 1e3, 3),
                 'tflops': round(v[0] / v[1] / 1e12, 1), 'share_of_step_time': round(v[1] / steps / step_s, 4)}
                for k, v in groups.items() if v[1] > 0]

    BIG = ops.LARGE_TILE_KERNELS          # every kernel template the router can pick for a tile of 128 rows or more (tests/test_host_cpu.py)

    def roofline_of(a, clips_per_s, tr, steps, step_s):
        D, L, _ = VIT[a.model]
        fclip = flops_per_clip(D, L, a.frames)
        step_tf = clips_per_s / world * fclip / 1e12
        out = {'achieved': round(step_tf, 1), 'frac': round(step_tf / MFMA_PEAK_TFLOPS, 4),
               'executed_frac': round(clips_per_s / world * (fclip - flops_skipped_per_clip(D, a.frames)) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
        if tr:
            rows = [r for r in gemm_rows(tr, steps, step_s) if r['kernel'].startswith(BIG)]
            fl = sum(r['tflops'] * r['ms_per_step'] for r in rows)            # TFLOP/s x ms = GFLOP per step
            tm = sum(r['ms_per_step'] for r in rows)
            if tm > 0:
                large = [r for r in rows if r['share_of_step_time'] >= 0.02]
                worst = min(large, key=lambda r: r['tflops']) if large else None
                out['gemm_family'] = {'tflops': round(fl / tm, 1), 'frac': round(fl / tm / MFMA_PEAK_TFLOPS, 4), 'ms_per_step': round(tm, 2),
                                      'share_of_step_time': round(tm * 1e-3 / step_s, 3)}
                if worst is not None:
                    out['worst_large_gemm_row'] = dict(worst, frac=round(worst['tflops'] / MFMA_PEAK_TFLOPS, 4),
                                                       note='lowest-rate (kernel template, shape) group among those taking >= 2 % of the step')
                out['rows'] = sorted(rows, key=lambda r: -r['ms_per_step'])[:8]
        return out

    if rank == 0:
        D, L, _ = VIT[args.model]
        fclip = flops_per_clip(D, L, args.frames)
        clips = args.batch * world * args.steps / elapsed
        step_s = elapsed / args.steps
        per_variant = {}
        if trace:
            for name, fl, e0, e1, _shape in trace:
                d = per_variant.setdefault(name, [0.0, 0.0, 0])
                d[0] += fl
                d[1] += e0.elapsed_time(e1) * 1e-3
                d[2] += 1
        dom = {k: v for k, v in per_variant.items() if k.startswith(BIG)}
        fl = sum(v[0] for v in dom.values())
        tm = sum(v[1] for v in dom.values())
        n_launch = sum(v[2] for v in dom.values())
        step_tf = clips / world * fclip / 1e12
        pmc = pmc_traffic(args)
        tsteps = max(m['trace_steps'], 1)
        tstep_s = m['trace_elapsed'] / tsteps if trace else step_s
        rl = roofline_of(args, clips, trace, tsteps, tstep_s)
        roof = {'bound': 'mfma', 'scope': 'whole training step (fwd + bwd + SGD) per GPU: clips/s/GPU x algorithmic GFLOP/clip (SURVEY 8d)',
                'achieved': round(step_tf, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(step_tf / MFMA_PEAK_TFLOPS, 4),
                'frac_counts': 'algorithmic FLOPs (SURVEY 8d), including the part of the last ViT block that the CLS-only evaluation skips',
                'executed_frac': rl['executed_frac'],
                'traffic': None if pmc is None else pmc['hbm_bytes_per_step'],
                'traffic_source': None if pmc is None else pmc.get('source'),
                'traffic_note': 'traffic is NOT measured in this run: it is the fabric read + write bytes per step of a committed rocprofv3 --pmc pass (FETCH_SIZE / WRITE_SIZE, '
                                'separate passes, gfx950 corrections) over the same workload -- traffic_source names the file; algorithmic_bytes_per_step is the per-kernel operand '
                                'model of bench.py::algorithmic_bytes_per_step (every tensor crossing HBM as often as the kernel sequence consumes / produces it), not SURVEY 8d\'s '
                                'secondary list of parameter / optimizer / input bytes (~13.5 GB, which it contains)',
                'algorithmic_bytes_per_step': algorithmic_bytes_per_step(D, L, args.frames, args.batch),
                'hbm_time_floor_ms_at_6p3TBs': round(algorithmic_bytes_per_step(D, L, args.frames, args.batch) / 6.3e12 * 1e3, 2),
                'peak_note': 'peak = dense bf16 MFMA at the nominal clock (MI355X_MICROARCH.md); a bare MFMA loop on pseudo-random bf16 '
                             'operands sustains 1.79-1.97 PFLOP/s on this part (power-limited clock, profiles/r02_mfma_feed_lab.txt); this bench '
                             'draws 1.35-1.37 kW of the 1.4 kW package cap at 1.84-1.89 GHz (profiles/r02_power_clock_under_bench.txt)'}
        if tm > 0:
            ach = fl / tm / 1e12
            roof['dominant_kernel'] = {
                'kernel': 'gemm_8pp_kernel<*> (persistent 256x256x64, 8-phase) + gemm_8p_kernel<*> (256x256x64, 8-phase) + gemm_w4_kernel (weight gradients, 4 waves of 128x128) + gemm_kernel<256|128,*> (bf16 MFMA GEMM, all layouts/epilogues)',
                'achieved': round(ach, 1), 'unit': 'TFLOP/s', 'frac': round(ach / MFMA_PEAK_TFLOPS, 4),
                'launches': n_launch, 'avg_launch_us': round(tm / n_launch * 1e6, 2),
                'avg_launch_gflop': round(fl / n_launch / 1e9, 3), 'share_of_step_time': round(tm / m['trace_elapsed'], 3),
                'ms_per_step': round(tm / tsteps * 1e3, 2),
                'measured_in': f"{tsteps} steps run after the timed region with a HIP-event pair around every GEMM launch ({round(tstep_s * 1e3, 2)} ms/step with the events)",
                'per_variant_tflops': {k: round(v[0] / v[1] / 1e12, 1) for k, v in per_variant.items() if v[1] > 0}}
            if 'worst_large_gemm_row' in rl:
                roof['worst_large_gemm_row'] = rl['worst_large_gemm_row']
                roof['largest_gemm_rows'] = rl['rows']
        name = 'ViT-B/16' if args.model.startswith('vit_base') else args.model
        out = {'metric': f'training clips/sec ({name}+AVT-h, {args.frames}x224^2 frames)',
               'value': round(clips, 2), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic' if args.backend == 'nccl' else 'synthetic (gloo functional check: ranks share devices, NOT a measurement)',
               'config': {'workload': f'{args.model} + AVT-h(2048x6x4) fwd+bwd+SGD-nesterov, T={args.frames} x 224^2, C={NUM_CLASSES}, '
                                      f'{args.batch} clips/GPU, dropout 0.1/0.2 on, fp32 master weights / bf16 MFMA',
                          'clips_per_gpu': args.batch, 'global_batch': args.batch * world, 'frames': args.frames,
                          'parallelism': f'dp{world}', 'gflop_per_clip': round(fclip / 1e9, 2), 'final_loss': round(loss_val, 4),
                          'launch': 'hipGraph replay (avt_amd/func/graph.py: device-resident dropout seeds and learning rates)' if (args.graph and world == 1) else 'eager'},
               'per_rank_clips_per_s': [round(args.batch * args.steps / float(x), 2) for x in per_rank],
               'host': {'abi_calls_per_step': round(abi_calls / args.steps, 1), 'enqueue_ms_per_step': round(host_enqueue / args.steps * 1e3, 2),
                        'note': 'rank 0: C-ABI calls (1-2 kernel launches each) and Python time to enqueue one step; enqueue >= ms_per_step means the step is host-bound'},
               'roofline': roof}
        if comm_all[0] is not None:
            out['comm'] = {'rccl_ranks_seen': ranks_seen, 'backend': args.backend, 'devices': devices_seen, 'per_rank': comm_all, 'note': 'comm_exposed_ms = time the optimizer waited for the gradient exchange after backward had finished',
                           'env': {k: v for k, v in os.environ.items() if k.startswith(('NCCL_', 'RCCL_', 'HSA_ENABLE_IPC'))}}
            if m['no_comm_trace'] and tm > 0:
                nc = [r for r in gemm_rows(m['no_comm_trace'], 3, step_s) if r['kernel'].startswith(BIG)]
                nc_ms = sum(r['ms_per_step'] for r in nc)
                out['comm']['gemm_family_ms_per_step'] = {'with_collectives_in_flight': round(tm / tsteps * 1e3, 2), 'exchange_paused': round(nc_ms, 2),
                                                          'note': 'rank 0, HIP events around every large-tile GEMM launch; the difference is what RCCL\'s kernels cost the GEMMs (CUs / HBM)'}

    # BASELINE configs 4 and 5 at their own architecture, short runs inside the same line (N = 1 only; --no-also skips them): half of the
    # headline's clips per GPU at T = 15 (128 of 256: the same 1920 frames), three eighths for ViT-L (96)
    if world == 1 and not args.no_also and (args.model, args.frames) == ('vit_base_patch16_224', 10):
        also = []
        for label, kw in (('config 4: ViT-B/16 + AVT-h, T = 15', dict(frames=15, batch=max(1, args.batch // 2))),
                          ('config 5: ViT-L/16 + AVT-h, T = 10', dict(model='vit_large_patch16_224', batch=max(1, args.batch * 3 // 8))),
                          # SURVEY 8d's batch list for config 2 ends at 64 clips per GPU (the reference's own runs use 3): the same model at a quarter of the headline's clips
                          ('config 2 at a quarter of the clips: ViT-B/16 + AVT-h, T = 10', dict(batch=max(1, args.batch // 4))),
                          # ... and at the batch the reference itself trains with (expts/01_ek100_avt.txt:5: 3 clips per GPU) -- weight-sized work (SGD, weight-gradient
                          # slabs, the head's weights) is most of that step
                          ('config 2 at the reference\'s own batch: ViT-B/16 + AVT-h, T = 10', dict(batch=3)),
                          # ... and that step replayed from a hipGraph (round 6, ABI 9: device-resident dropout seeds and learning rates): at 3 clips the host needs
                          # about as long to issue the ~640 launches as the device to run them
                          ('config 2 at the reference\'s own batch, the step replayed from a hipGraph: ViT-B/16 + AVT-h, T = 10', dict(batch=3, graph=True, no_gemm_trace=True))):
            a2 = argparse.Namespace(**{**vars(args), **kw})
            st2, wu2 = (20, 5) if a2.batch <= 8 and a2.batch < args.batch else (5, 2)      # (a 14-ms step needs more of them to leave its warm-up behind)
            try:
                m2 = measure(a2, st2, wu2)
            except Exception as e:                # the headline number must survive a failure of the extra runs (e.g. a box with less free HBM)
                also.append({'config': f'{label}, {a2.batch} clips/GPU', 'error': f'{type(e).__name__}: {e}'[:300]})
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                continue
            c2 = a2.batch * st2 / m2['elapsed_local']
            r2 = roofline_of(a2, c2, m2['trace'], max(m2['trace_steps'], 1), m2['trace_elapsed'] / max(m2['trace_steps'], 1) if m2['trace'] else m2['elapsed_local'] / st2)
            entry = {'config': f'{label}, {a2.batch} clips/GPU', 'model': a2.model, 'frames': a2.frames, 'clips_per_gpu': a2.batch,
                     'value': round(c2, 2), 'unit': 'clips/s', 'ms_per_step': round(m2['elapsed_local'] / st2 * 1e3, 3), 'steps': st2, 'warmup': wu2,
                     'frac': r2['frac'], 'executed_frac': r2['executed_frac'], 'final_loss': round(m2['loss'], 4),
                     'launch': 'hipGraph replay' if getattr(a2, 'graph', False) else 'eager', 'host_enqueue_ms_per_step': round(m2['host_enqueue'] / st2 * 1e3, 2)}
            if 'gemm_family' in r2:
                entry['gemm_family_frac'] = r2['gemm_family']['frac']
            if 'worst_large_gemm_row' in r2:
                w = r2['worst_large_gemm_row']
                entry['worst_large_gemm_row'] = {'kernel': w['kernel'], 'MNK': w['MNK'], 'tflops': w['tflops'], 'share_of_step_time': w['share_of_step_time']}
            also.append(entry)
        if rank == 0:
            out['also'] = also

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
