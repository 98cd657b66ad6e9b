"""-m gpu: the real data-parallel step (arena + backward segment hooks + bucketed all-reduce + fused SGD grad_scale) on
two ranks sharing cuda:0 over gloo (RCCL needs one device per rank; the 1-GPU box has one), against a single-process
step on the concatenated batch: averaged-gradient training must give the same parameters."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
root = sys.argv[1]; sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from helpers import build_hip_model, LOSS_WTS
from avt_amd.config import Cfg
from avt_amd.func.train import Trainer
from avt_amd.func.train_eval_ops import Basic
from avt_amd.optim import FusedSGD
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
backend = os.environ.get('AVT_TEST_BACKEND', 'gloo')
if backend == 'nccl' or os.environ.get('AVT_TEST_TRANSPORT') == 'abi':                      # RCCL: one device per rank
    assert torch.cuda.device_count() >= world
    torch.cuda.set_device(rank)
if world > 1:
    dist.init_process_group(backend, init_method='env://', rank=rank, world_size=world)      # (transport 'abi': only the side channel for the 128-byte id)
    assert dist.get_backend() == backend
torch.manual_seed(0)                       # identical init on every rank (broadcast must be a no-op then)
model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32), device=torch.device('cuda', torch.cuda.current_device()))
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.ndim >= 2: p.normal_(0, 0.1)
if rank == 1:                              # perturb: broadcast from rank 0 must repair it
    with torch.no_grad(): model.classifiers.action.bias.add_(1.0)
opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-4, arena=model.arena)
op = Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
# early optimizer step (round 6): AVT_TEST_EARLY_MIN=1 lets every exchanged bucket of this small model be stepped behind its collective; '0' switches it off
if os.environ.get('AVT_TEST_EARLY_MIN') == '0': Trainer.EARLY_STEP = False
elif os.environ.get('AVT_TEST_EARLY_MIN'): Trainer.EARLY_STEP, Trainer.EARLY_MIN_ELEMS = True, int(os.environ['AVT_TEST_EARLY_MIN'])      # (an option, off by default)
tr = Trainer(model, op, opt, None, LOSS_WTS, distributed=world > 1, bucket_bytes=64 << 10, reduce_mode=os.environ.get('AVT_TEST_REDUCE_MODE', 'all_reduce'),
             reduce_transport=os.environ.get('AVT_TEST_TRANSPORT', 'torch'))
early_calls = []
if tr.early_step:
    real_suffix = opt.step_suffix
    opt.step_suffix = lambda lo: (early_calls.append(lo), real_suffix(lo))[1]
g = torch.Generator().manual_seed(9)
B = int(os.environ.get('AVT_TEST_CLIPS', 4))
video = torch.rand((B, 4, 3, 1, 32, 32), generator=g) * 2 - 1
target = torch.randint(0, 17, (B,), generator=g); sub = torch.randint(-1, 17, (B, 4, 1), generator=g)
sl = slice(rank * B // world, (rank + 1) * B // world)
data = {'video': video[sl].cuda(), 'target': {'action': target[sl].cuda()}, 'target_subclips': {'action': sub[sl].cuda()}}
model.train()
for m in model.modules():                   # deterministic parity: dropout off
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
model.future_predictor.embd_pdrop = model.future_predictor.attn_pdrop = model.future_predictor.resid_pdrop = 0.0
for _ in range(3):
    tr.step(data)
torch.cuda.synchronize()
if rank == 0:
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, sys.argv[2])
if os.environ.get('AVT_TEST_EARLY_MIN', '0') not in ('0', ''):
    assert tr.early_step and len(early_calls) > 3 * 2 and (world == 1 or min(early_calls) == 0), early_calls      # several suffixes per step; data-parallel: finish() hands over the rest
if world > 1:
    assert tr.reducer is not None and tr.reducer.launched >= 1
    if os.environ.get('AVT_TEST_CHECK_COMM'):
        # world-size arithmetic of the exchange (bucket edges on 64 * world elements, at most one early tail flush, every gradient byte sent exactly once,
        # every rank ends on the same parameters bit for bit)
        r, a = tr.reducer, model.arena
        st = r.stats()
        assert r.bucket_elems % (64 * world) == 0 and r.bucket_elems == (64 << 10) // 4 // (64 * world) * (64 * world), r.bucket_elems
        assert st['bytes_per_step'] == a.total * 4, (st, a.total)
        full = a.total // r.bucket_elems
        assert full - 1 <= st['buckets_per_step'] <= full + 2, (st, full, r._tail_sent)      # full buckets (+ the early tail flush) + finish()'s head
        assert st['mode'] == os.environ.get('AVT_TEST_REDUCE_MODE', 'all_reduce') and st['comm_exposed_ms'] >= 0.0, st
        hi, lo = a.master.detach().clone(), a.master.detach().clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        assert torch.equal(hi, lo), 'the replicas drifted apart'
    if backend == 'nccl':                  # the ranks RCCL connected, counted on the devices
        one = torch.ones(1, device='cuda'); dist.all_reduce(one); assert int(one.item()) == world
    dist.barrier(); dist.destroy_process_group()
print('OK', rank)
'''


def _run(world, out, tmp_path, port, **extra_env):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(out)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for p in procs:
        o = p.communicate(timeout=600)[0]
        assert p.returncode == 0 and 'OK' in o, o[-3000:]


def test_two_rank_training_matches_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    _run(1, tmp_path / 'single.pt', tmp_path, 29551)
    _run(2, tmp_path / 'ddp.pt', tmp_path, 29552)
    a, b = torch.load(tmp_path / 'single.pt'), torch.load(tmp_path / 'ddp.pt')
    worst = 0.0
    for k in a:
        e = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-12))
        worst = max(worst, e)
        assert e < 2e-2, (k, e)          # bf16 activations; batch split changes rounding, not the maths


def test_early_optimizer_step_behind_each_bucket_matches_the_late_step(tmp_path):
    """Round 6: the fused SGD runs on every suffix of the gradient buffer as soon as its exchange is enqueued (on the exchange's stream, behind the
    collective) instead of in one pass after backward.  Two ranks, every 64-KiB bucket stepped on its own (AVT_TEST_EARLY_MIN=1), against the same job
    with the feature off: identical parameters, bit for bit (the update of an element does not depend on when it is applied)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    _run(2, tmp_path / 'late.pt', tmp_path, 29561, AVT_TEST_EARLY_MIN='0')
    _run(2, tmp_path / 'early.pt', tmp_path, 29562, AVT_TEST_EARLY_MIN='1')
    _run(1, tmp_path / 'early1.pt', tmp_path, 29563, AVT_TEST_EARLY_MIN='1')          # single process: the side-stream form
    _run(1, tmp_path / 'late1.pt', tmp_path, 29564, AVT_TEST_EARLY_MIN='0')
    for x, y in (('late.pt', 'early.pt'), ('late1.pt', 'early1.pt')):
        a, b = torch.load(tmp_path / x), torch.load(tmp_path / y)
        for k in a:
            assert torch.equal(a[k], b[k]), (x, y, k)


@pytest.mark.parametrize('mode', ['all_reduce', 'rs_ag'])
def test_eight_rank_training_on_one_gpu_matches_single_process(tmp_path, mode):
    """BASELINE config 3's world size without its node: EIGHT ranks of the real Trainer share cuda:0 over gloo, 2 clips per rank, both exchange forms --
    the parameters after three steps equal the single-process run on the 16 clips, and every rank checks the exchange's world-8 arithmetic itself
    (AVT_TEST_CHECK_COMM in the worker: bucket edges on 64 * 8 elements, the bucket count, arena.total * 4 payload bytes, identical replicas).
    The first RCCL run on a real node then cannot fail on world-size arithmetic (func/train.py:771-778)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    _run(1, tmp_path / 'single.pt', tmp_path, 29581, AVT_TEST_CLIPS='16')
    _run(8, tmp_path / 'ddp8.pt', tmp_path, 29582 + (mode == 'rs_ag'), AVT_TEST_CLIPS='16', AVT_TEST_REDUCE_MODE=mode, AVT_TEST_CHECK_COMM='1')
    a, b = torch.load(tmp_path / 'single.pt'), torch.load(tmp_path / 'ddp8.pt')
    for k in a:
        e = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-12))
        assert e < 2e-2, (k, e, mode)


def test_bench_eight_ranks_end_to_end_over_gloo():
    """`python bench.py --gpus 8 --backend gloo --batch 2`: the driver's 8-GPU command line with the ranks sharing the one device -- ONE JSON line with
    n_gpus = 8, global batch 16, dp8, eight per-rank rates and eight exchange records.  A functional check, not a measurement."""
    import json
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--backend', 'gloo', '--batch', '2', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 16 and d['config']['parallelism'] == 'dp8' and len(d['per_rank_clips_per_s']) == 8
    assert d['value'] > 0 and d['scaling'] == 'weak' and d['comm']['rccl_ranks_seen'] == 8 and len(d['comm']['per_rank']) == 8
    total = d['comm']['per_rank'][0]['bytes_per_step']
    assert all(c['buckets_per_step'] >= 1 and c['bytes_per_step'] == total for c in d['comm']['per_rank'])


def _devices():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.parametrize('mode', ['all_reduce', 'rs_ag'])
@pytest.mark.parametrize('ranks', ['2', 'all'])
def test_rccl_training_on_one_device_per_rank_matches_single_process(tmp_path, mode, ranks):
    """The replacement of the reference's DistributedDataParallel wrap (func/train.py:771-778, common/utils.py:145-148) over RCCL proper:
    one process per GPU, backend 'nccl', 2 ranks and as many ranks as the node has devices, both exchange forms -- three training steps on a
    split batch must give the parameters of a single-process run on the whole batch (same limits as the gloo test).  Skipped on a one-GPU
    box (the driver's 1-GPU tier); turns into RCCL parity evidence by itself on any node with two or more devices."""
    n = _devices()
    if n < 2:
        pytest.skip(f'RCCL needs one device per rank: {n} device(s) here')
    world = 2 if ranks == '2' else n
    if ranks == 'all' and n == 2:
        pytest.skip('covered by the 2-rank case')
    clips = str(2 * world)
    _run(1, tmp_path / 'single.pt', tmp_path, 29571, AVT_TEST_CLIPS=clips)
    _run(world, tmp_path / 'rccl.pt', tmp_path, 29572 + (mode == 'rs_ag'), AVT_TEST_BACKEND='nccl', AVT_TEST_REDUCE_MODE=mode, AVT_TEST_CLIPS=clips)
    a, b = torch.load(tmp_path / 'single.pt'), torch.load(tmp_path / 'rccl.pt')
    for k in a:
        e = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-12))
        assert e < 2e-2, (k, e, world, mode)


@pytest.mark.parametrize('mode', ['all_reduce', 'rs_ag'])
def test_abi_rccl_training_on_one_device_per_rank_matches_single_process(tmp_path, mode):
    """The same as above with the collectives issued through the C ABI (GradReducer(transport='abi'): avt_comm_init_rank on every rank from
    the id rank 0 drew, avt_broadcast_bucket, avt_allreduce_bucket | avt_reduce_scatter_bucket + avt_allgather_bucket); torch.distributed (gloo)
    only carries the 128-byte id.  Skipped on a one-GPU box; RCCL parity evidence for the ABI on any node with two or more devices."""
    n = _devices()
    if n < 2:
        pytest.skip(f'RCCL needs one device per rank: {n} device(s) here')
    clips = str(2 * n)
    _run(1, tmp_path / 'single.pt', tmp_path, 29591, AVT_TEST_CLIPS=clips)
    _run(n, tmp_path / 'abi.pt', tmp_path, 29592 + (mode == 'rs_ag'), AVT_TEST_TRANSPORT='abi', AVT_TEST_REDUCE_MODE=mode, AVT_TEST_CLIPS=clips)
    a, b = torch.load(tmp_path / 'single.pt'), torch.load(tmp_path / 'abi.pt')
    for k in a:
        e = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-12))
        assert e < 2e-2, (k, e, n, mode)


def test_rccl_collectives_through_the_c_abi_on_one_gpu():
    """The communicator and every collective of include/avt_hip.h's "gradient exchange over RCCL" section at world size 1 (a sum over one rank is the
    identity): fp32 and bf16, reduce-scatter + all-gather, broadcast, the size query, the host-side argument checks."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from avt_amd.comm import RcclComm
    from avt_amd.lib import AvtHipError
    c = RcclComm(1, 0, 0, RcclComm.unique_id())
    assert c.size() == (1, 0)
    g = torch.Generator(device='cuda').manual_seed(3)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(4096, device='cuda', generator=g).to(dt)
        y = x.clone()
        c.all_reduce(y); c.reduce_scatter(y); c.all_gather(y); c.broadcast(y, 0)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                                   # explicit stream: the call lands on the caller's current stream
        z = torch.ones(1024, device='cuda')
        c.all_reduce(z)
    s.synchronize()
    assert float(z.sum()) == 1024.0
    with pytest.raises(AvtHipError, match='fp32 / bf16'):
        c.all_reduce(torch.ones(8, device='cuda', dtype=torch.float16))
    with pytest.raises(AvtHipError, match='bad rank'):
        RcclComm(1, 3, 0, RcclComm.unique_id())
    c.destroy()


def test_bench_multi_gpu_line_over_rccl():
    """`python bench.py --gpus N` over RCCL on every device of the node (N >= 2): one JSON line, n_gpus = N, RCCL saw N ranks on N distinct
    devices, per-rank rates and the exchange accounting present.  Skipped on a one-GPU box."""
    import json
    n = _devices()
    if n < 2:
        pytest.skip(f'needs two or more devices: {n} here')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--batch', '8', '--steps', '3', '--warmup', '2',
                        '--no-cpu-baseline'], capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == n and d['comm']['rccl_ranks_seen'] == n and d['comm']['backend'] == 'nccl'
    assert len(set(d['comm']['devices'])) == n and len(d['per_rank_clips_per_s']) == n and d['value'] > 0
    assert all(c['buckets_per_step'] >= 1 for c in d['comm']['per_rank'])


NCCL_WORKER = r'''
import os, sys, torch, torch.distributed as dist
root = sys.argv[1]; sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from helpers import build_hip_model, LOSS_WTS
from avt_amd.config import Cfg
from avt_amd.common import utils
from avt_amd.ddp import GradReducer
from avt_amd.func.train import Trainer
from avt_amd.func.train_eval_ops import Basic
from avt_amd.optim import FusedSGD
on, rank, world, local = utils.init_distributed_mode('nccl', allow_single=True)
assert on and dist.get_backend() == 'nccl' and world == 1
torch.manual_seed(0)
model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True, arena=model.arena)
op = Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
tr = Trainer(model, op, opt, None, LOSS_WTS, distributed=True, bucket_bytes=64 << 10, force_reducer=True, reduce_mode=os.environ.get('AVT_TEST_REDUCE_MODE', 'all_reduce'),
             wire_dtype=torch.bfloat16 if os.environ.get('AVT_TEST_WIRE') == 'bf16' else torch.float32, reduce_transport=os.environ.get('AVT_TEST_TRANSPORT', 'torch'))
assert tr.reducer is not None and tr.reducer.transport == os.environ.get('AVT_TEST_TRANSPORT', 'torch')
if tr.reducer.transport == 'abi':
    assert tr.reducer.comm.size() == (1, 0)                  # asked of RCCL itself (ncclCommCount / ncclCommUserRank)
g = torch.Generator().manual_seed(9)
data = {'video': (torch.rand((2, 4, 3, 1, 32, 32), generator=g) * 2 - 1).cuda(), 'target': {'action': torch.randint(0, 17, (2,), generator=g).cuda()},
        'target_subclips': {'action': torch.randint(-1, 17, (2, 4, 1), generator=g).cuda()}}
before = model.classifiers.action.weight.detach().clone()
for _ in range(2):
    loss, _, _, _ = tr.step(data)
torch.cuda.synchronize()
assert tr.reducer.launched > 2, tr.reducer.launched          # bucketed RCCL all-reduces really ran, on the side stream
assert not torch.equal(before, model.classifiers.action.weight) and float(loss) == float(loss)
st = tr.reducer.stats()
assert st['buckets_per_step'] == tr.reducer.launched and st['comm_exposed_ms'] >= 0.0 and st['mode'] == os.environ.get('AVT_TEST_REDUCE_MODE', 'all_reduce'), st
assert st['bytes_per_step'] == model.arena.total * (2 if os.environ.get('AVT_TEST_WIRE') == 'bf16' else 4), st
dist.barrier(); dist.destroy_process_group()
print('OK nccl', tr.reducer.launched)
'''


@pytest.mark.parametrize('transport', ['torch', 'abi'])
@pytest.mark.parametrize('mode,wire', [('all_reduce', 'fp32'), ('rs_ag', 'fp32'), ('all_reduce', 'bf16'), ('rs_ag', 'bf16')])
def test_rccl_executes_the_bucketed_allreduce_on_one_gpu(tmp_path, mode, wire, transport):
    """backend='nccl' (= RCCL) with world_size 1: init_process_group, rank-0 broadcast, the bucketed all_reduce launched from
    the backward segment hooks on the side stream, finish() -- the same code path the 8-GPU run takes, on the box we have.
    transport 'abi': the same exchange through the library's own RCCL entry points (avt_comm_init_rank, avt_broadcast_bucket, avt_allreduce_bucket /
    avt_reduce_scatter_bucket + avt_allgather_bucket; include/avt_hip.h ABI 8) instead of torch.distributed's."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    script = tmp_path / 'nccl_worker.py'
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29561', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0', AVT_TEST_REDUCE_MODE=mode, AVT_TEST_WIRE=wire, AVT_TEST_TRANSPORT=transport)
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'OK nccl' in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` spawns its own ranks; with fewer devices than N it must fail loudly, not fall back."""
    n = torch.cuda.device_count() + 1 if torch.cuda.is_available() else 2
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(max(n, 2)), '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and 'refusing to oversubscribe' in (p.stdout + p.stderr)


def test_bench_two_ranks_end_to_end_over_gloo():
    """The N > 1 path of bench.py end to end on the box we have: `python bench.py --gpus 2` spawns its own ranks (torch.distributed.run
    on 127.0.0.1), both ranks train on cuda:0 over gloo (RCCL needs one device per rank), rank 0 prints ONE JSON line with
    n_gpus = 2, the whole-job rate and both per-rank rates.  A functional check, not a measurement."""
    import json
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--batch', '2', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['config']['parallelism'] == 'dp2' and len(d['per_rank_clips_per_s']) == 2
    assert d['value'] > 0 and 'cpu_baseline' not in d
    assert d['comm']['rccl_ranks_seen'] == 2 and d['comm']['backend'] == 'gloo' and len(d['comm']['devices']) == 2
    assert len(d['comm']['per_rank']) == 2 and all(c['buckets_per_step'] >= 1 and 'comm_exposed_ms' in c for c in d['comm']['per_rank'])
    assert d['host']['abi_calls_per_step'] > 100 and d['roofline']['executed_frac'] < d['roofline']['frac']
    # round 4: the line says what the collectives cost the GEMMs (same step with the exchange paused) and which RCCL knobs were in effect
    g = d['comm']['gemm_family_ms_per_step']
    assert g['with_collectives_in_flight'] > 0 and g['exchange_paused'] > 0 and 'env' in d['comm'] and 'also' not in d


def test_bench_single_gpu_line_carries_configs_4_and_5():
    """`python bench.py` at N = 1 appends short runs of BASELINE configs 4 (T = 15) and 5 (ViT-L/16) and of config 2 at two smaller batches to the ONE JSON line (`also`), and
    names the worst large GEMM row next to the family number.  Tiny batch here: a functional check of the line's shape."""
    import json
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', '8', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['config']['clips_per_gpu'] == 8 and d['value'] > 0
    also = d['also']
    # (configs 4 and 5 at half / three eighths of the clips, config 2 at a quarter of them and at the reference's own 3 clips per GPU)
    assert [a['frames'] for a in also] == [15, 10, 10, 10, 10]
    assert [a['model'] for a in also] == ['vit_base_patch16_224', 'vit_large_patch16_224', 'vit_base_patch16_224', 'vit_base_patch16_224', 'vit_base_patch16_224']
    assert [a['clips_per_gpu'] for a in also] == [4, 3, 2, 3, 3] and all(a['value'] > 0 and 0 < a['executed_frac'] < a['frac'] for a in also)
    # (round 6) the last entry is the reference's own batch again, the step replayed from a hipGraph
    assert [a['launch'] for a in also] == ['eager'] * 4 + ['hipGraph replay'] and all(a['host_enqueue_ms_per_step'] > 0 for a in also)
    w = d['roofline']['worst_large_gemm_row']
    assert w['tflops'] > 0 and len(w['MNK']) == 3 and w['share_of_step_time'] >= 0.02
