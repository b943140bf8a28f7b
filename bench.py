#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

metric : point-clouds/sec, forward+backward(+Adam step), PVCNN (1xC) S3DIS, N=4096, R=32, fp32,
         B=16 clouds per GPU (BASELINE configs[1]); weak scaling over N GPUs (one process per GPU,
         RCCL all-reduce of gradients, launched by torch.distributed.run).
A "step" = zero_grad -> forward -> cross-entropy -> backward (+ gradient all-reduce) -> Adam.step
on one synthetic batch already resident in HBM.

How the step is issued.  A PVCNN step is ~320 kernel launches of 4-250 us; one Python thread issues them slower (~25 us each)
than the GPU retires them, so an eagerly issued step measures the host, not the GPU.  The timed region therefore REPLAYS the
step from a hipGraph captured once (pvcnn_amd/graph.py: the same kernels on the same stream in the same order; 1 GPU: the
whole step incl. the fused Adam update; N GPUs: the whole step as well -- the bucketed RCCL all-reduces are captured where the
reducer's autograd hooks launch them, overlapped with the rest of backward -- or, if RCCL refuses the capture, forward + backward
captured and the all-reduces + Adam issued after each replay; `config.step_issue` says which).  `value` / `ms_per_step` are that region; `eager_value` is the same step issued launch by launch from
Python (timed separately, after the timed region).  `--eager` times the eager step instead.

Extra objects on the JSON line:
  roofline     : the hot path's headline kernel (trilinear_devoxelize fwd at the R=32 stage,
                 16x64x4096 points from a 16x64x32^3 grid), timed twice: LIVE in this process with HIP events on the
                 launch stream around every launch of it in instrumented eager steps AFTER the timed region (a replayed
                 graph cannot carry per-kernel events; median event-pair time - the calibrated time of an empty event pair
                 on a busy stream), and from the committed rocprofv3 trace of this command, the average of the SAME launch
                 inside the replayed graph (profiles/kernel_durations.json).  achieved = SURVEY 8(d) algorithmic bytes /
                 the SLOWER of the two; both are on the line.
  roofline_mfma: the step's largest MFMA-bound launch (Conv3d forward, 64->64 at 32^3), same two timings, same pricing,
                 against the dense fp16 MFMA peak (f16x2 executes 3 fp16 products per fp32 product).
  kernels      : the same for every watched kernel family that ran in the step.
  cpu_baseline : the same network on the host cores with the CPU oracle as native backend
                 (kind "port": the reference has no CPU implementation), bounded sample.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (RCCL / cross-process tensors fail otherwise)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch
import torch.distributed as dist
import torch.nn.functional as tf

MFMA_FP32_PEAK_TF = 157.3      # v_mfma_f32_32x32x2_f32, dense, MI355X_MICROARCH.md
MFMA_BF16_PEAK_TF = 2500.0     # v_mfma_f32_32x32x16_bf16, dense (not the 2:1-sparsity headline), MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable copy)


# ---- algorithmic bytes per call (the figure roofline.achieved is computed from) ----
def bytes_vox_fwd(b, c, n, s):
    return 4 * b * (c * n + 3 * n + c * s + n + s)


def bytes_vox_bwd(b, c, n, s):
    return 4 * b * (c * min(n, s) + n + min(n, s) + c * n)


def bytes_devox_fwd(b, c, n, s, training=True):
    return 4 * b * (3 * n + c * min(s, 8 * n) + c * n) + (64 * b * n if training else 0)


def bytes_devox_bwd(b, c, n, s):
    return 4 * b * (c * n + 16 * n + c * s)


class KernelClock:
    """Times native calls with HIP events recorded on the stream they are launched on (torch's
    current stream -- the stream handle pvcnn_amd passes through the C ABI)."""

    WATCH = {
        'avg_voxelize_forward': lambda a, out: ('avg_voxelize_fwd', bytes_vox_fwd(a[0].shape[0], a[0].shape[1], a[0].shape[2], int(a[2]) ** 3),
                                                (a[0].shape[0], a[0].shape[1], a[0].shape[2], int(a[2]))),
        'avg_voxelize_backward': lambda a, out: ('avg_voxelize_bwd', bytes_vox_bwd(a[0].shape[0], a[0].shape[1], a[1].shape[1], a[0].shape[2]),
                                                 (a[0].shape[0], a[0].shape[1], a[1].shape[1], round(a[0].shape[2] ** (1 / 3)))),
        'trilinear_devoxelize_forward': lambda a, out: ('trilinear_devoxelize_fwd',
                                                        bytes_devox_fwd(a[3].shape[0], a[3].shape[1], a[2].shape[2], int(a[0]) ** 3, bool(a[1])),
                                                        (a[3].shape[0], a[3].shape[1], a[2].shape[2], int(a[0]))),
        # the same gather with BatchNorm + LeakyReLU applied while the grid is staged (same argument positions)
        # (+ the point-branch addend read when PVConv's sum rides on the store: 4*B*C*N more compulsory bytes)
        'trilinear_devoxelize_bnact_forward': lambda a, out: ('trilinear_devoxelize_fwd',
                                                              bytes_devox_fwd(a[3].shape[0], a[3].shape[1], a[2].shape[2], int(a[0]) ** 3, bool(a[1]))
                                                              + (4 * a[9].numel() if len(a) > 9 and a[9] is not None else 0),
                                                              (a[3].shape[0], a[3].shape[1], a[2].shape[2], int(a[0]))),
        'trilinear_devoxelize_backward': lambda a, out: ('trilinear_devoxelize_bwd',
                                                         bytes_devox_bwd(a[0].shape[0], a[0].shape[1], a[0].shape[2], int(a[3]) ** 3),
                                                         (a[0].shape[0], a[0].shape[1], a[0].shape[2], int(a[3]))),
    }

    # the scatters run as plan (once per (coords, R), shared by the layers) + apply (per layer): both are timed; the
    # algorithmic bytes of SURVEY 8(d) are charged to the apply, the plan is listed with the bytes it must touch
    WATCH.update({
        'avg_voxelize_apply': lambda a, out: ('avg_voxelize_fwd', bytes_vox_fwd(a[0].shape[0], a[0].shape[1], a[0].shape[2], a[1].r ** 3),
                                              (a[0].shape[0], a[0].shape[1], a[0].shape[2], a[1].r)),
        'avg_voxelize_plan': lambda a, out: ('avg_voxelize_plan (counting sort, shared by the layers at this R)',
                                             4 * a[0].shape[0] * (4 * a[0].shape[2] + int(a[1]) ** 3), (a[0].shape[0], 0, a[0].shape[2], int(a[1]))),
        'trilinear_devoxelize_backward_apply': lambda a, out: ('trilinear_devoxelize_bwd',
                                                               bytes_devox_bwd(a[0].shape[0], a[0].shape[1], a[0].shape[2], int(a[2]) ** 3),
                                                               (a[0].shape[0], a[0].shape[1], a[0].shape[2], int(a[2]))),
        'trilinear_devoxelize_backward_plan': lambda a, out: ('trilinear_devoxelize_bwd_plan (counting sort, shared by the layers at this R)',
                                                              4 * a[0].shape[0] * 16 * a[0].shape[2], (a[0].shape[0], 0, a[0].shape[2], int(a[2]))),
        # (round 5) both of the above from one launch chain, built when the first layer voxelizes a (coords, R): the bytes of the two
        'pvconv_plans': lambda a, out: ('avg_voxelize_plan + trilinear_devoxelize_bwd_plan in ONE chain (shared by the layers at this R)',
                                        4 * a[0].shape[0] * (4 * a[0].shape[2] + int(a[2]) ** 3) + 4 * a[0].shape[0] * 16 * a[0].shape[2],
                                        (a[0].shape[0], 0, a[0].shape[2], int(a[2]))),
    })

    # MFMA-bound family: the Conv3d implicit-GEMM launches (forward and backward-data are the same kernel); the "bytes" slot
    # carries the ALGORITHMIC FLOPs 2*B*R^3*27*Ci*Co.  conv3d_igemm_split(x, wts, bias, co, nsplit): the launch alone (the
    # weight image and, for f16x2, the input's amax buffer are prepared outside the timed call); nsplit = 3 / 2 executes 6 bf16 / 3 fp16 MFMA
    # products per algorithmic one.
    WATCH_FLOPS = {
        'conv3d_forward': lambda a, out: ('conv3d_igemm fp32-MFMA (incl. weight transform)',
                                          2.0 * a[0].shape[0] * a[0].shape[2] ** 3 * 27 * a[0].shape[1] * a[1].shape[0],
                                          (a[0].shape[0], a[0].shape[1], a[1].shape[0], a[0].shape[2])),
        'conv3d_igemm_split': lambda a, out: ('conv3d_igemm_bf16 ' + {3: 'bf16x3', 2: 'f16x2', 1: 'bf16'}[int(a[4])],
                                              2.0 * a[0].shape[0] * a[0].shape[2] ** 3 * 27 * a[0].shape[1] * int(a[3]),
                                              (a[0].shape[0], a[0].shape[1], int(a[3]), a[0].shape[2])),
    }

    def __init__(self, backend):
        self.backend, self.records, self.enabled, self._orig, self.only = backend, [], False, {}, None
        self.last_call = {}                     # (kernel, shape) -> (callable, args, kw) of its last watched call: burst()

    def install(self):
        for name, describe in list(self.WATCH.items()) + list(self.WATCH_FLOPS.items()):
            orig = getattr(self.backend, name)
            self._orig[name] = orig

            def timed(*args, _orig=orig, _describe=describe, **kw):
                if not self.enabled:
                    return _orig(*args, **kw)
                if self.only is not None:            # timed region: only the two roofline kernels carry an event pair
                    key = _describe(args, None)
                    if (key[0], tuple(key[2])) not in self.only:
                        return _orig(*args, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _orig(*args, **kw)
                e1.record()
                key = _describe(args, out)
                self.records.append((key, e0, e1))
                if key[0] == 'trilinear_devoxelize_fwd':      # (the roofline kernel's family: burst() replays its last launch)
                    self.last_call[(key[0], tuple(key[2]))] = (_orig, args, kw)
                return out
            setattr(self.backend, name, timed)

    def uninstall(self):
        for name in self._orig:
            try:
                delattr(self.backend, name)      # drop the instance attribute -> class method again
            except AttributeError:
                pass

    @staticmethod
    def event_pair_overhead_us(pairs=200):
        """What an (event, event) pair with NOTHING in between reads on a busy stream: the part of every
        measurement below that is not the kernel.  The pairs are queued behind a long kernel so that, as in
        the timed steps, the GPU consumes them back to back instead of waiting for the host."""
        filler = torch.randn(8192, 8192, device='cuda')
        for _ in range(3):                       # ~30 ms of queued work: far longer than recording the pairs takes
            filler = (filler @ filler) * 1e-4
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(pairs)]
        for e0, e1 in evs:
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
        return t[len(t) // 2]

    def burst(self, kernel, shape, launches=64):
        """The last watched call of (kernel, shape) again, `launches` times back to back behind a long filler kernel, an event pair
        around each (median; the caller subtracts event_pair_overhead_us, calibrated the same way): us per launch with
        the GPU -- not the Python thread that feeds the stream -- setting the pace.  An event pair around a single launch inside an
        eager step reads the HOST's gap whenever the host is the slower side: one evidence run of round 5 read 62 us (median of 60
        pairs) for the kernel rocprofv3 saw at 35.9 us inside the replayed graph on the same box (profiles/r05_bench_20_5.json as
        committed in 24f205d)."""
        call = self.last_call.get((kernel, tuple(shape)))
        if call is None:
            return None
        fn, args, kw = call
        for _ in range(4):
            fn(*args, **kw)
        filler = torch.randn(8192, 8192, device='cuda')
        for _ in range(3):                       # ~30 ms of queued work: the launches below are all queued before the first one starts
            filler = (filler @ filler) * 1e-4
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for e0, e1 in evs:
            e0.record()
            fn(*args, **kw)
            e1.record()
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
        return t[len(t) // 2]

    def summary(self, overhead_us=0.0):
        # MEDIAN of the event pairs of a launch shape (round 5): one host hiccup between the two records of a pair -- the stream is
        # fed launch by launch from Python here -- used to move the mean by 10 us (a 47.9 us "live" reading of a 35.9 us kernel in one
        # evidence run), and the roofline is priced on the slower of this figure and the in-graph one
        import statistics
        samples = {}
        for (kernel, nbytes, shape), e0, e1 in self.records:
            samples.setdefault((kernel, shape), [nbytes, []])[1].append(e0.elapsed_time(e1))
        agg = {k: [len(v[1]), statistics.median(v[1]) * len(v[1]), v[0]] for k, v in samples.items()}
        out = []
        for (kernel, shape), (calls, ms, nbytes) in sorted(agg.items()):
            raw_us = ms * 1e3 / calls
            us = max(raw_us - overhead_us, 1e-3)
            if kernel.startswith('conv3d_igemm'):       # MFMA-bound
                tf_s = nbytes / (us * 1e-6) / 1e12              # algorithmic ("effective fp32") rate
                mult = 6 if 'bf16x3' in kernel else 3 if 'f16x2' in kernel else 1    # MFMA products executed per algorithmic product
                peak = MFMA_FP32_PEAK_TF if 'fp32' in kernel else MFMA_BF16_PEAK_TF
                out.append({'kernel': kernel, 'shape_BCiCoR': list(shape), 'calls': calls, 'avg_us': round(us, 2),
                            'event_pair_us': round(raw_us, 2), 'GFLOP': round(nbytes / 1e9, 2),
                            'effective_TFLOPs': round(tf_s, 1), 'executed_mfma_TFLOPs': round(tf_s * mult, 1), 'peak_TFLOPs': peak,
                            'frac_of_peak': round(tf_s * mult / peak, 4),
                            'x_fp32_mfma_peak': round(tf_s / MFMA_FP32_PEAK_TF, 3)})
                continue
            gbs = nbytes / (us * 1e-6) / 1e9
            out.append({'kernel': kernel, 'shape_BCNR': list(shape), 'calls': calls, 'avg_us': round(us, 2),
                        'event_pair_us': round(raw_us, 2),
                        'algorithmic_MB': round(nbytes / 1e6, 3), 'achieved_GBs': round(gbs, 1),
                        'frac_of_8TBs': round(gbs / HBM_PEAK_GBS, 4)})
        return out


def price_launch_us(burst_us, in_step_us, graph_us):
    """Which of the three timings of the roofline launch `achieved` / `frac` are priced on -> (us, label).  ALWAYS the slower of two
    (ADVICE r04): the GPU-paced burst of this run and the rocprofv3 in-graph average -- the latter only while the committed trace was
    taken on the kernel sources this run executes (`graph_us` is None otherwise: a stale trace never prices a run).  The burst
    re-reads one grid 64 times, the 256 MB memory-side cache serves part of it: it UNDER-reads the launch of the step, whose grid the
    convolution in front has just written.  So without an in-graph figure the host-paced in-step event pair -- an upper bound --
    prices the line with it."""
    if graph_us:
        us = max(burst_us, graph_us)
        return us, ('rocprofv3 in-graph average (profiles/kernel_durations*.json)' if us == graph_us else 'live HIP events of this run')
    us = max(burst_us, in_step_us)
    return us, ('live HIP events of this run (in-step pairs: no trace of the running kernel sources)' if us == in_step_us and in_step_us != burst_us
                else 'live HIP events of this run')


def dtype_label(backend):
    """fp32 tensors and fp32 accumulation everywhere; what differs is how the dense-convolution PRODUCTS are formed."""
    if getattr(backend, 'conv_math', 'fp32') == 'f16x2':
        pw = getattr(backend, 'pw_math', 'fp32') == 'f16x2'
        return ('f32 (fp32 tensors, fp32 accumulate; Conv3d fwd / bwd-data / bwd-weight' + (' and the large SharedMLP fwd / bwd-data / bwd-weight' if pw else '')
                + ' products as power-of-two scaled fp16 hi+lo splits, 3 partial products on fp16 MFMA, max rel err vs fp64 1e-6 <= the '
                'fp32-MFMA kernels\'; PVCNN_CONV_MATH=fp32 PVCNN_PW_MATH=fp32 select single-rounding fp32 MFMA, =bf16x3 the scale-free '
                '6-product split)')
    if getattr(backend, 'conv_math', 'fp32') == 'bf16x3':
        return ('f32 (fp32 tensors, fp32 accumulate; Conv3d fwd/bwd-data products as exact 3-way bf16 splits on bf16 MFMA, '
                'max rel err vs fp64 2e-6 = the fp32-MFMA kernel\'s; PVCNN_CONV_MATH=fp32 selects single-rounding fp32 MFMA)')
    return 'f32'


def pmc_traffic(kernel, shape):
    """HBM bytes per launch of `kernel` at `shape` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    written by tools/pmc_by_kernel.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of THIS bench command;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide streaming reads on gfx950).  Nothing is hard-coded:
    no file, or no entry for this kernel and shape -> traffic null."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return {'traffic': None}
    from pvcnn_amd._lib import sources_digest
    if table.get('sources_digest') != sources_digest():        # counters of OTHER kernel sources say nothing about this run
        return {'traffic': None, 'traffic_note': f"profiles/pmc_traffic.json was collected on kernel sources {table.get('sources_digest')}, "
                                                 f"this run executes {sources_digest()}: not cited"}
    for row in table.get('kernels', []):
        if row.get('op') == kernel and list(row.get('shape_BCNR', [])) == list(shape):
            out = {'traffic': int((2 * row['FETCH_SIZE_KiB'] + row['WRITE_SIZE_KiB']) * 1024),
                   'traffic_source': f"profiles/pmc_traffic.json: {row.get('kernel_name', '?')}, FETCH_SIZE {row['FETCH_SIZE_KiB']} KiB "
                                     f"(x2, gfx950 correction) + WRITE_SIZE {row['WRITE_SIZE_KiB']} KiB; {table.get('command', '')}"}
            if row.get('note'):
                out['traffic_note'] = row['note']
            return out
    return {'traffic': None}


def trace_table(config='cfg2'):
    """The committed per-launch table of this command's rocprofv3 trace + whether it was taken on the kernel sources this run executes."""
    path = os.path.join(ROOT, 'profiles', 'kernel_durations.json' if config == 'cfg2' else f'kernel_durations_{config}.json')
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None, {'file': os.path.relpath(path, ROOT), 'present': False}
    from pvcnn_amd._lib import sources_digest
    now = sources_digest()
    return table, {'file': os.path.relpath(path, ROOT), 'present': True, 'trace_commit': table.get('trace_commit'),
                   'trace_sources_digest': table.get('sources_digest'), 'running_sources_digest': now,
                   'matches_running_sources': table.get('sources_digest') == now}


def in_graph_us(needles, nth=None, config='cfg2'):
    """Average duration of one launch of a kernel INSIDE the replayed step graph, from the committed rocprofv3 kernel trace of this very
    command (profiles/kernel_durations.json -- kernel_durations_cfgN.json for the other configs --, written by tools/trace_steady.py
    --json; one entry per launch of a step: kernel name, grid, position among the equal launches).  -> (avg_us, kernel name) of the
    SLOWEST matching launch, or (None, None): no file, a file taken on OTHER kernel sources than the ones running (digest mismatch: a
    stale trace must not price this run), or no launch whose name holds all `needles`."""
    table, info = trace_table(config)
    if table is None or not info['matches_running_sources']:
        return None, None
    hits = [r for r in table.get('launches', []) if all(n in r['kernel'] for n in needles) and (nth is None or r['nth_in_step'] == nth)]
    if not hits:
        return None, None
    hits.sort(key=lambda r: -r['avg_us'])
    return hits[0]['avg_us'], hits[0]['kernel'][:110]


def run_variant(config, extra_args=(), env=None, steps=20, warmup=5, timeout=300):
    """The same bench command in a process of its own (the arithmetic switches are read once per process; a second model in this
    process would share the weight bank and the allocator pools with the first): -> its JSON line, or {'error': ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--config', config, '--steps', str(steps), '--warmup', str(warmup),
           '--no-cpu-baseline', '--no-variants', *extra_args]
    try:
        out = subprocess.run(cmd, env={**os.environ, **(env or {})}, capture_output=True, text=True, timeout=timeout)
        lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
        if out.returncode != 0 or not lines:
            return {'error': f'rc={out.returncode}: {out.stderr.strip()[-300:]}'}
        return json.loads(lines[-1])
    except Exception as exc:                                   # noqa: BLE001 -- reported on the line
        return {'error': f'{type(exc).__name__}: {str(exc)[:200]}'}


def _dry_rank(rank, world, port, queue):
    """One rank of `--dry-collectives`: gloo over CPU tensors, a small pure-torch network with several gradient buckets, the two
    sequencings pvcnn_amd/graph.py captures on the GPU issued eagerly (capture=False) -- the collectives are the REAL ones of
    pvcnn_amd/dp.py in the order a replayed graph would issue them."""
    import torch.nn as nn
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pvcnn_amd.dp import GradBucketReducer, shard_batch
        from pvcnn_amd.graph import GraphedTrainStep
        out = {}
        gen = torch.Generator().manual_seed(7)
        x = torch.randn(4 * world, 6, 32, generator=gen)
        y = torch.randint(0, 5, (4 * world, 32), generator=gen)
        sl = shard_batch(4 * world, world, rank)
        for name, overlapped in (('graph+collectives (hooks launch each full bucket during backward)', True),
                                 ('graph, collectives after replay (fixed bucket order)', False)):
            torch.manual_seed(100 + rank)                       # different initial weights per rank: the broadcast must fix that
            model = nn.Sequential(nn.Conv1d(6, 24, 1), nn.GroupNorm(4, 24), nn.ReLU(), nn.Conv1d(24, 24, 1), nn.ReLU(), nn.Conv1d(24, 5, 1))
            reducer = GradBucketReducer(model, bucket_mb=0.0005)
            opt = torch.optim.SGD(model.parameters(), lr=0.1)
            step = GraphedTrainStep(model, lambda: tf.cross_entropy(model(x[sl]), y[sl]), opt, reducer, capture=False, capture_collectives=overlapped)
            start = [p.detach().numpy().copy() for p in model.parameters()]
            losses = [float(step().detach()) for _ in range(3)]
            out[name] = {'buckets': len(reducer.buckets), 'start': start, 'end': [p.detach().numpy().copy() for p in model.parameters()], 'losses': losses}
        queue.put((rank, out))
    finally:
        dist.destroy_process_group()


def dry_collectives(world):
    """`bench.py --gpus N --dry-collectives` (no GPU): N gloo ranks run the multi-rank step sequencings; rank results are compared with
    each other (lock-step) and with big-batch SGD in one process.  -> the JSON line (not a performance figure)."""
    import socket
    import torch.multiprocessing as mp
    import torch.nn as nn
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context('spawn')
    queue = ctx.Queue()
    procs = [ctx.Process(target=_dry_rank, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(queue.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    report = {'dry_collectives': True, 'backend': 'gloo (CPU tensors)', 'ranks': world, 'exit_codes': [p.exitcode for p in procs], 'orderings': {}}
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(4 * world, 6, 32, generator=gen)
    y = torch.randint(0, 5, (4 * world, 32), generator=gen)
    for name, r0 in results[0].items():
        lock_step = all(all((a == b).all() for a, b in zip(r0['end'], results[r][name]['end'])) for r in range(1, world))
        model = nn.Sequential(nn.Conv1d(6, 24, 1), nn.GroupNorm(4, 24), nn.ReLU(), nn.Conv1d(24, 24, 1), nn.ReLU(), nn.Conv1d(24, 5, 1))
        with torch.no_grad():
            for p, src in zip(model.parameters(), r0['start']):
                p.copy_(torch.from_numpy(src))
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        for _ in range(3):                                      # (no batch statistics in this stand-in: DP == big-batch SGD exactly)
            opt.zero_grad()
            tf.cross_entropy(model(x), y).backward()
            opt.step()
        worst = max(float((torch.from_numpy(a) - b.detach()).abs().max()) for a, b in zip(r0['end'], model.parameters()))
        report['orderings'][name] = {'gradient_buckets': r0['buckets'], 'ranks_in_lock_step': lock_step,
                                     'max_abs_diff_vs_big_batch_sgd': worst, 'ok': bool(lock_step and worst < 1e-5)}
    report['ok'] = all(v['ok'] for v in report['orderings'].values()) and all(c == 0 for c in report['exit_codes'])
    return report


def cpu_baseline(args, sample_batch):
    """The same network + step on the host CPU with the oracle as native backend (kind "port")."""
    from oracle.oracle_backend import OracleBackend          # checker / baseline only
    from pvcnn_amd import workload
    from pvcnn_amd.modules.functional import backend as seam
    # threads actually used: torch intra-op pool capped at 32 (beyond that the small per-layer GEMMs of
    # a B=2 step only pay fork/join overhead; on a 256-core host 256 threads ran 80x SLOWER)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    hip = seam._backend
    seam._backend = OracleBackend()
    try:
        torch.manual_seed(workload.SEED)
        model = workload.PVCNN(13, 6, width_multiplier=args.width).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        x, y = workload.make_s3dis_batch(sample_batch, args.points)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = tf.cross_entropy(model(x), y)
            loss.backward()
            opt.step()
        step()                                               # warm-up
        t0, n = time.perf_counter(), 0
        while True:
            step()
            n += 1
            el = time.perf_counter() - t0
            if el > 20.0 or n >= 5:
                break
    finally:
        seam._backend = hip
    return {'value': round(sample_batch * n / el, 3), 'unit': 'point-clouds/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} fwd+bwd+Adam steps of PVCNN {args.width}xC at B={sample_batch}, N={args.points} '
                      f'(oracle C backend + torch-CPU conv/BN, {cores} threads; 1 warm-up step excluded)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--batch', type=int, default=0, help='clouds per GPU (default: cfg2 16, cfg3 8, cfg4 8, cfg5 32)')
    ap.add_argument('--points', type=int, default=0)
    ap.add_argument('--width', type=float, default=1.0)
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg3', 'cfg4', 'cfg5'],
                    help='BASELINE.json configs: cfg2 PVCNN S3DIS (the headline, default); cfg3 PVCNN++ S3DIS B=8 N=8192; cfg4 PVCNN ShapeNet '
                         'B=8/GPU N=2048; cfg5 Frustum-PVCNN KITTI B=4/GPU... (per-GPU batches of the 8-GPU runs unless --batch is given)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-batch', type=int, default=0, help='clouds per CPU-baseline step (0 = the same batch as the GPU run)')
    ap.add_argument('--bucket-mb', type=float, default=8.0)
    ap.add_argument('--eager', action='store_true',
                    help='time the eagerly issued step (one Python thread issues every launch: host-bound) instead of the hipGraph replay')
    ap.add_argument('--graph', action='store_true', help='(default since round 3; accepted for older command lines)')
    ap.add_argument('--collectives-after-replay', action='store_true',
                    help='N > 1: do not capture the RCCL all-reduces inside the graph; issue them (and Adam) after each replay')
    ap.add_argument('--single-rank-collectives', action='store_true',
                    help='self-test of the multi-GPU step on ONE GPU: a 1-rank RCCL process group, the reducer forced to issue its bucket '
                         'all-reduces (the identity over one rank) -- the timed region is then the N > 1 code path (graph + captured '
                         'collectives) with everything but a second rank')
    ap.add_argument('--torch-adam', action='store_true', help='torch.optim.Adam(fused=True) instead of pvcnn_amd.optim.FlatAdam')
    ap.add_argument('--reference-composition', action='store_true',
                    help='time the network composed EXACTLY as the reference\'s models/ compose it (tests/reference_composition.py: plain '
                         '.max / .repeat / torch.cat, nn.Sequential heads with nn.Dropout and nn.Conv1d as modules) on pvcnn_amd.modules -- '
                         'what a user gets who swaps the `modules` package and nothing else -- instead of pvcnn_amd.workload\'s composition')
    ap.add_argument('--no-variants', action='store_true',
                    help='do not add reference_composition_value / fp32_mfma_value to the line (each is this command again in a process of its own, 20 steps)')
    ap.add_argument('--dry-collectives', action='store_true',
                    help='no GPU needed: run the N-rank step SEQUENCING (both orderings of pvcnn_amd/graph.py\'s multi-rank branch, issued eagerly) '
                         'on a reduced-width network over gloo / CPU tensors with the CPU oracle as native backend and check it against the '
                         'single-process step on the concatenated batch; prints one JSON line (not a performance figure)')
    args = ap.parse_args()

    if args.dry_collectives:
        report = dry_collectives(max(2, args.gpus))
        print(json.dumps(report), flush=True)
        raise SystemExit(0 if report['ok'] else 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- one process per GPU under torch.distributed.run
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
                                  '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]])
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the PVConv hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    multi = world > 1 or args.single_rank_collectives           # the code path with a process group and collectives
    if multi and world == 1:
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            os.environ.setdefault('MASTER_PORT', str(sock.getsockname()[1]))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # whole-step graph capture with collectives inside (pvcnn_amd/graph.py): torch's documented set-up for captured NCCL work is to
        # switch the process group's asynchronous error handling off -- its watchdog thread otherwise polls events while a capture runs
        os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '0')
        dist.init_process_group('nccl', device_id=dev)       # backend "nccl" IS RCCL on ROCm

    from pvcnn_amd import workload
    from pvcnn_amd.dp import GradBucketReducer
    from pvcnn_amd.modules.functional import backend as seam
    assert seam._backend.name == 'hip-gfx950'
    seam._backend.lib                                         # dlopen now: fail loudly before timing

    # No MIOpen find-mode: the 3-D convolutions run on pvcnn_amd's own MFMA kernels, and an exhaustive
    # search for the remaining 1x1 convolutions would cost minutes of start-up for nothing.
    torch.backends.cudnn.benchmark = False
    torch.manual_seed(workload.SEED)
    defaults = {'cfg2': (16, 4096), 'cfg3': (8, 8192), 'cfg4': (8, 2048), 'cfg5': (32, 1024)}[args.config]
    args.batch = args.batch or defaults[0]
    args.points = args.points or defaults[1]
    autocast = contextlib.nullcontext
    composed = None
    if args.reference_composition:
        if args.config == 'cfg5':
            raise SystemExit('--reference-composition: cfg2 / cfg3 / cfg4 (tests/reference_composition.py)')
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import reference_composition                              # bench / test infrastructure: the reference's forward() on pvcnn_amd.modules
        composed = reference_composition.BY_CONFIG[args.config]
    if args.config == 'cfg2':      # BASELINE configs[1]: the headline
        model = workload.PVCNN(13, 6, width_multiplier=args.width)
        x, y = workload.make_s3dis_batch(args.batch, args.points, device=dev, seed=workload.SEED + rank)
        loss_of = lambda out: tf.cross_entropy(out, y)
        label = f'PVCNN ({args.width:g}xC) S3DIS fwd+bwd+Adam, B={args.batch}/GPU N={args.points} R=32/16 fp32'
        metric = 'point-clouds/sec fwd+bwd, PVCNN S3DIS N=4096 R=32'
    elif args.config == 'cfg3':    # configs[2]: ball_query + grouping + FPS + 3-NN path
        model = workload.PVCNN2(13, 6, width_multiplier=args.width)
        x, y = workload.make_s3dis_batch(args.batch, args.points, device=dev, seed=workload.SEED + rank)
        loss_of = lambda out: tf.cross_entropy(out, y)
        label = f'PVCNN++ ({args.width:g}xC) S3DIS fwd+bwd+Adam, B={args.batch}/GPU N={args.points} fp32'
        metric = 'point-clouds/sec fwd+bwd, PVCNN++ S3DIS N=8192'
    elif args.config == 'cfg4':    # configs[3]: SE blocks, normalize=False, one-hot concat
        model = workload.PVCNNShapeNet(50, 16, 3, width_multiplier=args.width)
        x, y = workload.make_shapenet_batch(args.batch, args.points, device=dev, seed=workload.SEED + rank)
        loss_of = lambda out: tf.cross_entropy(out, y)
        label = f'PVCNN ({args.width:g}xC) ShapeNet part-seg fwd+bwd+Adam, B={args.batch}/GPU N={args.points} R=32/16 fp32'
        metric = 'point-clouds/sec fwd+bwd, PVCNN ShapeNet N=2048 R=32'
    else:                          # configs[4]: Frustum-PVCNN, voxel convolutions on bf16 operands (torch.autocast)
        from pvcnn_amd.modules import FrustumPointNetLoss
        templates = workload.frustum_size_templates()
        model = workload.FrustumPVCNNE(3, 12, 8, 512, templates, 1, args.width)
        x, _ = workload.make_frustum_batch(args.batch, args.points, device=dev, seed=workload.SEED + rank)
        targets = workload.make_frustum_targets(args.batch, args.points, device=dev, seed=workload.SEED + rank)
        criterion = FrustumPointNetLoss(12, 8, templates).to(dev)
        loss_of = lambda out: criterion(out, targets)
        autocast = lambda: torch.autocast('cuda', dtype=torch.bfloat16)
        label = (f'Frustum-PVCNN ({args.width:g}xC) KITTI fwd+bwd+Adam, B={args.batch}/GPU N={args.points} R=16/12, torch.autocast(bf16): '
                 'Conv3d on bf16 MFMA operands, fp32 accumulate; device-side logits_mask')
        metric = 'frustums/sec fwd+bwd, Frustum-PVCNN KITTI N=1024'
    if composed is not None:       # same constructor (same parameters, same state_dict), the reference's own forward()
        torch.manual_seed(workload.SEED)
        model = composed[0](*composed[1], width_multiplier=args.width)
        label += ' -- composed as the reference\'s models/ (plain .max / .repeat / torch.cat, nn.Sequential heads with nn.Dropout, nn.Conv1d)'
    model = model.to(dev).train()
    reducer = GradBucketReducer(model, bucket_mb=args.bucket_mb, always_reduce=args.single_rank_collectives)
    # Adam (lr 1e-3, weight decay 1e-5: configs/s3dis/__init__.py) on the reducer's flat buckets: one elementwise launch per bucket,
    # step counter on the device (pvcnn_amd/optim.py; same arithmetic as torch.optim.Adam -- tests/test_gpu_optim.py).
    # --torch-adam: torch's fused multi-tensor Adam over the ~100 parameter tensors instead (2 x 72 us per PVCNN step).
    if args.torch_adam:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True, capturable=True)
    else:
        from pvcnn_amd.optim import FlatAdam
        opt = FlatAdam(reducer, lr=1e-3, weight_decay=1e-5)

    clock = KernelClock(seam._backend)
    clock.install()

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks(ok):
        """True iff `ok` on every rank (the ranks must take the same branch: a captured and an eager rank would deadlock)."""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    # ---- the step, captured once (pvcnn_amd/graph.py) ----
    from pvcnn_amd.graph import GraphedTrainStep
    graphed, graph_error = None, None
    if not args.eager:
        try:
            graphed = GraphedTrainStep(model, lambda: loss_of(model(x)), opt, reducer, autocast=autocast, warmup=3,
                                       capture_collectives=False if args.collectives_after_replay else None)
        except Exception as exc:                                 # reported on the line; the eager step is timed instead
            import traceback
            frames = [f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in traceback.extract_tb(exc.__traceback__)]
            graph_error = f'{type(exc).__name__}: {str(exc)[:160]} @ ' + ' < '.join(reversed(frames[-8:]))
        if not all_ranks(graphed is not None):
            graphed = None
            graph_error = graph_error or 'capture failed on another rank'

    def eager_step():
        reducer.zero_grad()
        with autocast():
            loss = loss_of(model(x))
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    step = graphed if graphed is not None else eager_step
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank_ms = [round(float(g.item()) / args.steps * 1e3, 3) for g in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.detach())

    # ---- the step's collectives alone (all buckets, back to back, nothing to overlap with): what NOT overlapping them would cost ----
    comm_us = None
    if multi:
        for _ in range(3):
            for b in reducer.buckets:
                dist.all_reduce(b.flat)
        fence()
        t0 = time.perf_counter()
        for _ in range(20):
            for b in reducer.buckets:
                dist.all_reduce(b.flat)
        fence()
        comm_us = round((time.perf_counter() - t0) / 20 * 1e6, 1)

    # ---- the same step issued launch by launch (host-bound on one Python thread): eager_value ----
    eager_steps = min(args.steps, 20)
    eager_elapsed = None
    if graphed is not None:
        # (the eager steps allocate from the default stream's pool, which is empty after a capture: the caching allocator needs a few
        # steps of its own before a step stops calling hipMalloc -- with 3 warm-up steps this figure read 1 100 ... 2 150 clouds/s from
        # process to process on the same code)
        for _ in range(10):
            eager_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(eager_steps):
            eager_step()
        fence()
        eager_elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([eager_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            eager_elapsed = t.item()

    # ---- per-kernel HIP events: instrumented eager steps (every watched launch carries an event pair on the launch stream) ----
    event_overhead_us = KernelClock.event_pair_overhead_us()
    clock.enabled = True
    for _ in range(eager_steps):
        eager_step()
    fence()
    clock.enabled = False
    kernels = clock.summary(event_overhead_us)
    clock.uninstall()

    if rank == 0:
        global_batch = args.batch * world
        head = next((k for k in kernels if k['kernel'] == 'trilinear_devoxelize_fwd' and k['shape_BCNR'][3] == max(
            kk['shape_BCNR'][3] for kk in kernels if kk['kernel'] == 'trilinear_devoxelize_fwd')), None)
        convs = [k for k in kernels if k['kernel'].startswith('conv3d_igemm')]
        mfma = max(convs, key=lambda k: (k['GFLOP'], k['calls'])) if convs else None
        roofline = None
        if head:
            b_, c_, n_, r_ = head['shape_BCNR']
            fused = args.config != 'cfg5'                       # under autocast the BatchNorm tail is not folded into the gather
            pipe = r_ == 32 and os.environ.get('PVCNN_GATHER_PIPE', '1') != '0'
            survey_bytes = bytes_devox_fwd(b_, c_, n_, r_ ** 3, True)
            # the same kernel inside the replayed graph by rocprofv3 (committed trace of this command): `frac` is priced on the SLOWER of
            # the two timings -- in the graph the gather starts right behind the convolution that wrote its grid
            # (other configs: the slowest devoxelize gather of the step IS the one at the largest resolution -- there the live figure is
            # an event pair on a host-bound stream around a 15-25 us launch and over-reads; the in-graph average is the kernel)
            # (the step's slowest devoxelize gather IS the one at the largest resolution; nothing about the shape is hard-coded)
            needles = ('gather_lds_pipe_kernel', 'TrilinearFromCoords', 'XfBnAct') if (pipe and fused and args.config == 'cfg2') else ('gather_lds', 'TrilinearFromCoords')
            graph_us, graph_kernel = in_graph_us(needles, config=args.config)
            # live figure: the step's own launch of this kernel (same arguments) replayed back to back with the GPU setting the pace;
            # the per-launch pairs inside the eager steps stay on the line as `in_step_event_pair_us` (host-paced: an upper bound)
            in_step_us = head['avg_us']
            burst_raw = clock.burst('trilinear_devoxelize_fwd', head['shape_BCNR'])
            if burst_raw is not None:
                head = dict(head, avg_us=round(max(burst_raw - event_overhead_us, 1e-3), 2), event_pair_us=round(burst_raw, 2))
                head['achieved_GBs'] = round(head['algorithmic_MB'] * 1e6 / (head['avg_us'] * 1e-6) / 1e9, 1)
            priced_us, priced_on = price_launch_us(head['avg_us'], in_step_us, graph_us)
            roofline = {'bound': 'hbm',
                        'kernel': ('pvcnn::gather_lds_pipe_kernel<TrilinearFromCoords, XfBnAct>' if pipe else 'pvcnn::gather_lds_kernel<TrilinearFromCoords>')
                                  + ' = trilinear_devoxelize fwd' + (' with PVConv\'s last BatchNorm+LeakyReLU applied in its LDS staging and the point branch added in its store' if fused else ''),
                        'shape_BCNR': head['shape_BCNR'],
                        'achieved': round(survey_bytes / (priced_us * 1e-6) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(survey_bytes / (priced_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        'priced_on_us': round(priced_us, 2),
                        'priced_on': priced_on,
                        'in_graph_us': graph_us, 'in_graph_kernel': graph_kernel, 'in_graph_trace': trace_table(args.config)[1],
                        'live_frac': round(survey_bytes / (head['avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        'algorithmic_MB': round(survey_bytes / 1e6, 3),
                        'algorithmic_bytes_formula': 'SURVEY 8(d): 4B(3N + C*min(S,8N) + C*N) + 64BN',
                        # the launch also reads the fused addend (4BCN more compulsory bytes): stated separately, not in `frac`
                        'with_fused_addend': {'algorithmic_MB': head['algorithmic_MB'], 'achieved': head['achieved_GBs'],
                                              'frac': round(head['achieved_GBs'] / HBM_PEAK_GBS, 4)},
                        'frac_of_achievable_6300': round(survey_bytes / (priced_us * 1e-6) / 1e9 / 6300.0, 4),
                        'avg_us': head['avg_us'], 'event_pair_us': head['event_pair_us'], 'calls': head['calls'],
                        'in_step_event_pair_us': in_step_us,
                        'event_overhead_us': round(event_overhead_us, 2),
                        'timing': 'avg_us: HIP events on the launch stream around each of 64 back-to-back launches of this kernel with the '
                                  'arguments of its last launch in the step, queued behind 30 ms of other work so that the GPU and not the '
                                  'Python thread sets the pace (MEDIAN event-pair time - the time an empty event pair reads under the same '
                                  f'conditions); in_step_event_pair_us: the same pairs around its launches inside {eager_steps} eager steps '
                                  '(host-paced: over-reads when the host is the slower side; a replayed graph cannot carry per-kernel '
                                  'events); in_graph_us: rocprofv3 average of the same kernel inside the replayed graph, from the committed '
                                  'trace of this command; achieved / frac use the SLOWER of avg_us and in_graph_us (of avg_us and in_step_event_pair_us '
                                  'when the committed trace is not of the running kernel sources)'}
            roofline.update(pmc_traffic('trilinear_devoxelize_fwd', head['shape_BCNR']))
        # the 64->64 forward at 32^3 is the SECOND launch of its template in a step (behind the 9->64 one, same grid): priced like the
        # HBM roofline on the slower of live events and the committed in-graph rocprofv3 average
        # (round 6: the 64 -> 64 forward at 32^3 is the FIRST launch of conv3d_igemm_f16_wide_kernel<32> in a step, its backward-data the
        #  second; under PVCNN_CONV_WIDE=0 it is the second launch of the two-workgroup template, behind the 9 -> 64 one)
        mfma_graph_us = None
        if mfma is not None and args.config == 'cfg2' and mfma['shape_BCiCoR'] == [16, 64, 64, 32]:
            mfma_graph_us = (in_graph_us(('conv3d_igemm_f16_wide_kernel<32',), nth=0)[0]
                             or in_graph_us(('conv3d_igemm_bf16_kernel<2, 4, 4, 32',), nth=1)[0])
        mfma_price = 1.0 if not (mfma and mfma_graph_us) else mfma['avg_us'] / max(mfma['avg_us'], mfma_graph_us)
        timed = 'hipGraph replay' if graphed is not None else 'eager'
        line = {
            'metric': metric,
            'value': round(global_batch * args.steps / elapsed, 2),
            'unit': 'frustums/s' if args.config == 'cfg5' else 'point-clouds/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('bf16 Conv3d operands (fp32 accumulate), fp32 elsewhere' if args.config == 'cfg5' else dtype_label(seam._backend)),
            'data': 'synthetic',
            'config': {'workload': label, 'baseline_config': args.config,
                       'composition': ('reference: tests/reference_composition.py' if composed is not None else
                                       'pvcnn_amd.workload (caller-side fusions around the operator API; see reference_composition_value)'),
                       'global_batch': global_batch, 'points': args.points, 'parallelism': f'dp{world}',
                       'gradient_bytes': reducer.gradient_bytes, 'final_loss': round(final_loss, 4),
                       'step_issue': (timed + {'graph': ' (1 GPU: zero_grad, forward, loss, backward, fused Adam in one graph launch)',
                                               'graph+collectives': ' of the whole step: the bucketed RCCL all-reduces are captured where the '
                                                                    'reducer\'s hooks launch them, overlapped with the rest of backward; fused Adam inside',
                                               'graph, collectives after replay': ' of forward + backward; bucketed RCCL all-reduce + fused Adam issued '
                                                                                  'after each replay (not overlapped)'}[graphed.mode]
                                      if graphed is not None else 'eager: one Python thread issues every launch; bucket all-reduces launched '
                                                                  'from the autograd hooks (overlapped with backward)'),
                       'step_mode': graphed.mode if graphed is not None else 'eager',
                       'rccl_ranks': world if multi else 0, 'per_rank_ms_per_step': per_rank_ms,
                       'gradient_buckets': len(reducer.buckets), 'allreduce_alone_us_per_step': comm_us},
            'timed_region': timed,
            'eager_value': None if eager_elapsed is None else round(global_batch * eager_steps / eager_elapsed, 2),
            'eager_ms_per_step': None if eager_elapsed is None else round(eager_elapsed / eager_steps * 1e3, 3),
            'roofline': roofline,
            # the step's largest MFMA-bound launch (Conv3d forward of the R=32 stage), same event timing
            'roofline_mfma': None if mfma is None else {
                'bound': 'mfma', 'kernel': mfma['kernel'] + ' (Conv3d 3x3x3 forward / backward-data of the largest stage)',
                'in_graph_us': mfma_graph_us, 'live_frac': mfma['frac_of_peak'],
                'shape_BCiCoR': mfma['shape_BCiCoR'], 'achieved': round(mfma['executed_mfma_TFLOPs'] * mfma_price, 1), 'peak': mfma['peak_TFLOPs'],
                'unit': 'TFLOP/s', 'frac': round(mfma['frac_of_peak'] * mfma_price, 4), 'avg_us': mfma['avg_us'], 'algorithmic_GFLOP': mfma['GFLOP'],
                'effective_fp32_TFLOPs': mfma['effective_TFLOPs'], 'x_fp32_mfma_peak_157TF': mfma['x_fp32_mfma_peak'],
                'note': 'achieved = MFMA flops actually executed (f16x2: 3 fp16 partial products per fp32 product; bf16x3: 6) / launch time; '
                        'effective = algorithmic 2*B*R^3*27*Ci*Co / launch time; frac is priced on the slower of live events and the '
                        'committed in-graph rocprofv3 average (in_graph_us).  frac = pipe busy x clock / 2.4 GHz: the counters of the '
                        'matrix kernels inside this step are profiles/r06_pmc_mfma_bench_table.md (busy share of the cycles, effective '
                        'clock), the same binaries on random and on constant operands profiles/r06_pmc_mfma_fill_probe.jsonl, and what '
                        'the parts of this kernel cost next to its bare MFMA stream profiles/r06_ablate_conv_fwd.jsonl (DESIGN 4)'},
            'kernels': kernels,
            'kernels_note': 'per-launch HIP event pairs inside the eager steps behind the timed region (median per launch shape, minus the '
                            'empty-pair time): host-paced, i.e. an upper bound per kernel; the in-graph durations are profiles/kernel_durations*.json',
        }
        if graph_error:
            line['graph_error'] = graph_error
        if graphed is not None and graphed.capture_error:
            line['collective_capture_error'] = graphed.capture_error
        if world == 1 and not args.no_variants and not args.reference_composition and not args.eager and args.config in ('cfg2', 'cfg3', 'cfg4'):
            # the same command twice more, each in a process of its own (20 timed steps): what a user of the reference gets who swaps
            # the `modules` package and keeps their models/ unchanged, and (cfg2) the single-rounding fp32-MFMA arithmetic
            passthrough = ['--batch', str(args.batch), '--points', str(args.points), '--width', str(args.width)]
            v = run_variant(args.config, ['--reference-composition', *passthrough])
            line['reference_composition_value'] = v.get('value')
            line['reference_composition'] = ({'value': v.get('value'), 'ms_per_step': v.get('ms_per_step'), 'timed_region': v.get('timed_region'),
                                              'eager_value': v.get('eager_value'), 'eager_ms_per_step': v.get('eager_ms_per_step'),
                                              'steps': v.get('steps'), 'what': 'this command with --reference-composition: the network composed as '
                                              'models/s3dis/pvcnn.py:34-46 + models/utils.py:15-46 compose it (tests/reference_composition.py), on pvcnn_amd.modules only'}
                                             if 'error' not in v else v)
            if args.config == 'cfg2':
                v = run_variant(args.config, passthrough, env={'PVCNN_CONV_MATH': 'fp32', 'PVCNN_PW_MATH': 'fp32'})
                line['fp32_mfma_value'] = v.get('value')
                line['fp32_mfma'] = ({'value': v.get('value'), 'ms_per_step': v.get('ms_per_step'), 'steps': v.get('steps'), 'dtype': v.get('dtype'),
                                      'what': 'this command under PVCNN_CONV_MATH=fp32 PVCNN_PW_MATH=fp32: every Conv3d / 1x1 product on '
                                              'v_mfma_f32_32x32x2_f32 (single-rounding fp32 MFMA) instead of the f16x2 split'}
                                     if 'error' not in v else v)
        if world == 1 and not args.no_cpu_baseline and args.config == 'cfg2':
            line['cpu_baseline'] = cpu_baseline(args, args.cpu_sample_batch or args.batch)
    # the JSON line must be the LAST line of the job's stdout: RCCL writes a start-up banner ("RCCL version : ...", five lines per rank)
    # through C stdio, which -- into a pipe -- stays in libc's buffer until the process exits, i.e. behind anything Python has printed.
    # Every rank flushes it, the ranks meet, THEN rank 0 prints.
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if multi:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
