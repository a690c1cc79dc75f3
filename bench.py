#!/usr/bin/env python
"""DenseASPP161 training throughput on MI355X: images/s at 1024x2048, data-parallel over N GPUs.

One step = im2col/stem ... classifier forward, fused bilinear-x8 + cross-entropy, full backward,
bucketed gradient all-reduce (N > 1) and the fused Adam update, on synthetic Cityscapes-shaped
batches already resident in HBM.  Prints ONE JSON line (rank 0).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (spec, without 2:1 sparsity)
PEAK_HBM_GBS = 8000.0         # MI355X HBM3E peak (spec)


def synth_batch(B, H, W, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, 3, H, W, generator=g).clamp_(-2.2, 2.7)
    labels = torch.randint(0, 19, (B, H, W), generator=g)
    labels[torch.rand(B, H, W, generator=g) < 0.1] = 255
    return x.to(device), labels.to(device)


def cpu_baseline(cfg, budget_s=40.0):
    """The oracle (plain torch.nn fp32 DenseASPP161) forward+backward on the host cores, batch 1: 3 warm-up and 5 timed
    iterations, median reported, at the largest of 1024x2048 / 512x1024 / 256x512 whose 8 iterations fit the budget."""
    import torch.nn.functional as F
    from oracle.dense_aspp_ref import DenseASPPRef
    threads = int(os.environ.get('OMP_NUM_THREADS', '0')) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    ref = DenseASPPRef(cfg).train()

    def one(H, W):
        x, labels = synth_batch(1, H, W, 0, 'cpu')
        t = time.time()
        ref.zero_grad()
        loss = F.cross_entropy(ref(x), labels, ignore_index=255)
        loss.backward()
        return time.time() - t

    one(128, 256)                       # thread-pool / allocator warm-up, untimed
    t_probe = one(256, 512)
    full = 1024 * 2048
    for H, W in ((1024, 2048), (512, 1024), (256, 512)):
        if 8 * t_probe * (H * W) / (256 * 512) <= budget_s or (H, W) == (256, 512):
            break
    for _ in range(3):
        one(H, W)
    times = sorted(one(H, W) for _ in range(5))
    t = times[2]
    # images/s in units of 1024x2048 images: the cost of this network is proportional to the pixel count
    return dict(value=(H * W / full) / t, unit='img/s', cores=threads, kind='port',
                sample='fp32 torch.nn oracle, fwd+bwd, batch 1 at %dx%d: 3 warm-up + 5 timed steps, median %.2f s '
                       '(min %.2f, max %.2f); value = 1024x2048-image equivalents per second (cost ~ pixels)'
                       % (H, W, t, times[0], times[-1]))


def reference_loop_rate(model, x, labels, steps=3, warmup=1):
    """The reference's own training loop (graph_seg_main.py:231-245) on the same model and batch:
    optimizer.zero_grad(); output = model(x); loss = CrossEntropyLoss(ignore_index=255)(output, gt); loss.backward();
    torch.optim.Adam.step() -- full-resolution fp32 logits are materialised and torch computes the loss on them."""
    crit = torch.nn.CrossEntropyLoss(ignore_index=255).to(x.device)
    opt = torch.optim.Adam(model.parameters(), weight_decay=1e-4, lr=1e-4)
    model.train()

    def one():
        opt.zero_grad()
        loss = crit(model(x), labels)
        loss.backward()
        opt.step()
        return loss

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    return dict(value=round(x.shape[0] / dt, 3), unit='img/s', ms_per_step=round(1e3 * dt, 2), batch=x.shape[0], steps=steps,
                loss=round(float(loss.detach()), 5),
                loop='zero_grad; model(x) -> fp32 NCHW logits; CrossEntropyLoss(ignore_index=255); backward(); torch.optim.Adam.step()')


def config4_rate(device, H, W, steps=3, warmup=1, reserve=0.15):
    """BASELINE config 4: DenseASPP201 training at 1024x2048 with the largest per-GPU batch that fits the HBM.  The
    batch is found from the memory one engine of 4 images takes: whole multiples of 8 images that leave `reserve` of the
    device memory free."""
    import importlib
    import denseASPP
    from semantic_segmentation_amd.optim import FusedAdam
    cfg = importlib.import_module('cfgs.DenseASPP201').Model_CFG
    torch.manual_seed(0)
    model = denseASPP.DenseASPP(cfg, n_class=19, output_stride=8).to(device).train()
    opt = FusedAdam(model, lr=1e-4, weight_decay=1e-4)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    model.engine(4, H, W, True)
    per_img = (torch.cuda.memory_allocated() - base) / 4.0
    model._engines.clear()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    fixed = 8 * 4 * sum(p.numel() for p in model.parameters())       # Adam moments + staging that appear with the first step
    B = int((free - reserve * total - fixed) / (per_img + 3 * 4 * H * W + 8 * H * W))
    B = max(8, min(B // 8 * 8, 96))
    x, labels = synth_batch(B, H, W, 0, device)
    gen = torch.Generator(device=device)
    gen.manual_seed(7)
    model.engine(B, H, W, True)
    model.refresh_weights()

    def one():
        loss = model.train_step(x, labels, generator=gen, refresh=False)
        opt.step()
        return loss

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    out = dict(model='DenseASPP201', batch=B, value=round(B / dt, 3), unit='img/s', ms_per_step=round(1e3 * dt, 2), steps=steps,
               loss=round(float(loss), 5), gb_per_image=round(per_img / 1e9, 3),
               gb_allocated=round(torch.cuda.max_memory_allocated() / 1e9, 1),
               mfma_peak_frac_whole_step=round(6.0 * model.conv_macs(H, W) * B / dt / 1e12 / PEAK_BF16_TFLOPS, 4))
    model._engines.clear()
    del model, opt, x, labels
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    ap.add_argument('--force-reducer', action='store_true',
                    help='one rank, but run the bucketed all-reduce path of N > 1 (rehearsal)')
    ap.add_argument('--comm-cus', type=int, default=None,
                    help='compute units the persistent kernels leave free while gradient buckets are in flight (default: '
                         'distributed.DEFAULT_COMM_CUS when collectives are issued)')
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--model', default='DenseASPP161')
    ap.add_argument('--side-wgrad', action='store_true',
                    help='dense-layer weight gradients on a second HIP stream (overlapped kernels: per-kernel times inflate)')
    ap.add_argument('--side-wgrad3', action='store_true', help='dense-layer 3x3 weight gradients on a second stream')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the two extra measurements of a one-GPU run: the reference-shaped loop and BASELINE config 4')
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that print to fd 1 (RCCL writes a version banner there when a
    # communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    # (rehearsal of the multi-rank control flow on a one-GPU box: DSASPP_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # exchanges through gloo with the device drained before each collective; it checks the rendezvous, the broadcast of
    # the model, the bucket bookkeeping and the reduction of the timings, and its numbers mean nothing)
    rehearsal = os.environ.get('DSASPP_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get('DSASPP_BENCH_WATCHDOG', '120')), exit=True)   # a hang names its line
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1 or args.force_reducer:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if rehearsal:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            # torch's gloo backend on this ROCm image never completes an asynchronous collective of a GPU tensor issued
            # behind queued kernels (tools/gloo_cuda_probe.py shows it with torch's own matmuls): drain the device first
            _all_reduce = dist.all_reduce

            def _drained_all_reduce(t, *a, **k):
                if t.is_cuda:
                    torch.cuda.synchronize()
                return _all_reduce(t, *a, **k)
            dist.all_reduce = _drained_all_reduce
        else:
            # RCCL runs one workgroup per channel; no more channels than the compute units the persistent kernels leave free
            # while gradient buckets are in flight, or the collective's workgroups and the one-block-per-CU kernels queue
            # behind each other (an explicit NCCL_MAX_NCHANNELS in the environment wins)
            import denseASPP  # noqa: F401  (registers the package under its importable name)
            from semantic_segmentation_amd.distributed import DEFAULT_COMM_CUS
            cus = args.comm_cus if args.comm_cus is not None else int(os.environ.get('DSASPP_COMM_CUS', DEFAULT_COMM_CUS))
            if cus > 0:
                os.environ.setdefault('NCCL_MAX_NCHANNELS', str(cus))
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    import importlib
    import denseASPP
    from semantic_segmentation_amd import profiling
    from semantic_segmentation_amd.distributed import GradReducer
    from semantic_segmentation_amd.optim import FusedAdam
    cfg = importlib.import_module('cfgs.' + args.model).Model_CFG

    torch.manual_seed(0)                                   # identical replicas on every rank
    model = denseASPP.DenseASPP(cfg, n_class=19, output_stride=8).to(device).train()
    model.side_wgrad = args.side_wgrad
    model.side_wgrad3 = args.side_wgrad3
    opt = FusedAdam(model, lr=1e-4, weight_decay=1e-4)     # reference defaults: Adam, coupled L2 1e-4
    B, H, W = args.batch, args.height, args.width
    x, labels = synth_batch(B, H, W, rank, device)
    gen = torch.Generator(device=device)
    gen.manual_seed(99 + rank)
    eng = model.engine(B, H, W, True)
    reducer = GradReducer(model, eng, always=args.force_reducer, comm_cus=args.comm_cus) if (world > 1 or args.force_reducer) else None
    model.refresh_weights()

    def step():
        t_s = time.time()
        loss = model.train_step(x, labels, generator=gen, refresh=False, reducer=reducer)
        opt.step(grad_scale=1.0 / world)
        if rehearsal:
            torch.cuda.synchronize()
            print('[rehearsal] rank %d step %.2f s' % (rank, time.time() - t_s), file=sys.stderr, flush=True)
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    sync()
    timing = rank == 0 and not args.no_kernel_timing
    progs = [eng._fwd, eng._fwd_loss, eng._bwd]
    nt = len(profiling.CLASSES)

    def tag(only=None):
        per = {}
        for p in progs:
            for t, (fl, by, n) in profiling.tag_program(p, only).items():
                a = per.setdefault(t, [0.0, 0.0, 0])
                a[0] += fl; a[1] += by; a[2] += n
        return per

    def read_timing():
        ms, cnt = [0.0] * nt, [0] * nt
        for p in progs:
            m, c = p.read_timing(nt)
            ms = [a + b for a, b in zip(ms, m)]
            cnt = [a + b for a, b in zip(cnt, c)]
            profiling.untag_program(p)
        return ms, cnt

    # Kernel table: ONE extra untimed step with every MFMA launch bracketed by HIP events (an event pair costs a few
    # microseconds of idle GPU, ~700 pairs a step).  The timed steps then bracket only the launches of the class that
    # took the most time in it, so that the measurement of that kernel perturbs the step as little as possible.
    kernels = {}
    dom_tag = None
    if timing:
        per_run = tag()
    if not args.no_kernel_timing:          # every rank takes the step (it holds the gradient exchange), rank 0 times it
        step()
        sync()
    if timing:
        ms, cnt = read_timing()
        for t in range(1, nt):
            if cnt[t]:
                fl, by, n = per_run[t]
                kernels[profiling.CLASSES[t]] = dict(
                    launches_per_step=n, avg_us=1e3 * ms[t] / cnt[t], ms_per_step=ms[t],
                    tflops=(fl / (ms[t] * 1e-3) / 1e12) if fl else None, gbs=(by / (ms[t] * 1e-3) / 1e9) if by else None)
        mfma = {k: v for k, v in kernels.items() if v['tflops']}
        dom = max(mfma, key=lambda k: mfma[k]['ms_per_step'])
        dom_tag = profiling.CLASSES.index(dom)
        dom_run = tag(dom_tag)[dom_tag]
    sync()
    t0 = time.time()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.time() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    loss_val = float(loss)
    if rehearsal and world > 1:
        # every rank saw other data: the replicas stay bit-identical only if every gradient went through the exchange
        chk = torch.stack([model._flat.double().sum(), model._flat.double().abs().sum()]).cpu()
        seen = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(seen, chk)
        same = all(torch.equal(seen[0], c) for c in seen)
        print('[rehearsal] rank %d: replicas identical after %d steps: %s' % (rank, args.warmup + args.steps + 1, same),
              file=sys.stderr, flush=True)
        if not same:
            raise SystemExit('rehearsal: the replicas diverged')

    roofline = None
    if timing:
        # the dominant class over the timed steps
        ms, cnt = read_timing()
        fl, by, n = dom_run
        t = dom_tag
        d = dict(launches_per_step=n, avg_us=1e3 * ms[t] / cnt[t], ms_per_step=ms[t] / args.steps,
                 tflops=fl * args.steps / (ms[t] * 1e-3) / 1e12, gbs=(by * args.steps / (ms[t] * 1e-3) / 1e9) if by else None)
        kernels[dom].update(d)
        # HBM bytes per launch from the committed PMC passes of this same build, per kernel class (bytes per launch scale
        # with the batch and change with the kernels: a record of another build or batch is not reported)
        traffic_tab = {}
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            from semantic_segmentation_amd import _lib
            tj = json.load(open(tpath))
            meta = tj.get('_meta') or {}
            if meta.get('batch') == args.batch and meta.get('source_hash') == _lib.source_hash():
                traffic_tab = {k: v.get('hbm_bytes_per_launch') for k, v in tj.items() if isinstance(v, dict) and k != '_meta'}
        for kname, kv in kernels.items():
            fl_k, by_k, n_k = per_run[profiling.CLASSES.index(kname)]
            kv['alg_bytes_per_launch'] = round(by_k / n_k) if n_k else None
            kv['traffic'] = traffic_tab.get(kname)
            kv['traffic_over_alg'] = round(kv['traffic'] / kv['alg_bytes_per_launch'], 3) if kv['traffic'] and kv['alg_bytes_per_launch'] else None
        traffic = traffic_tab.get(dom)
        # Every convolution is priced against the MFMA roof (dense contraction), whatever its algorithmic intensity says:
        # the basis on which the fractions of successive builds are comparable.  The HBM side of the same launches is
        # carried beside it (hbm_gbs / hbm_frac of the algorithmic bytes, flop_per_byte against the ridge of the machine,
        # and `binding`: which of the two roofs the algorithmic intensity puts the class under).
        intensity = d['tflops'] * 1e12 / (d['gbs'] * 1e9) if d.get('gbs') else float('inf')
        ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        roofline = dict(bound='mfma', achieved=round(d['tflops'], 2), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
                        frac=round(d['tflops'] / PEAK_BF16_TFLOPS, 4), kernel=dom, traffic=traffic,
                        alg_bytes_per_launch=kernels[dom].get('alg_bytes_per_launch'),
                        avg_launch_us=round(d['avg_us'], 2), launches_per_step=d['launches_per_step'],
                        flop_per_byte=round(intensity, 1), ridge_flop_per_byte=round(ridge, 1),
                        binding='hbm' if intensity < ridge else 'mfma',
                        mfma_tflops=round(d['tflops'], 2), mfma_frac=round(d['tflops'] / PEAK_BF16_TFLOPS, 4),
                        hbm_gbs=round(d['gbs'], 1) if d.get('gbs') else None,
                        hbm_frac=round(d['gbs'] / PEAK_HBM_GBS, 4) if d.get('gbs') else None)

    if rank == 0:
        imgs = world * B * args.steps
        macs = model.conv_macs(H, W)
        value = imgs / dt
        out = {
            'metric': 'DenseASPP161 train img/s @1024x2048' if args.model == 'DenseASPP161' and (H, W) == (1024, 2048)
                      else '%s train img/s @%dx%d' % (args.model, H, W),
            'value': round(value, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': '%s fwd+bwd+Adam, 19 classes, %dx%d, batch %d per GPU, bf16 storage / fp32 accumulate'
                                   % (args.model, H, W, B), 'global_batch': world * B, 'parallelism': 'dp%d' % world},
            'model_tflops_per_s': round(6.0 * macs * value / 1e12, 2),
            'mfma_peak_frac_whole_step': round(6.0 * macs * value / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
            'loss': round(loss_val, 5),
            'source_hash': sys.modules['semantic_segmentation_amd']._lib.source_hash(),
            'roofline': roofline,
            'kernels': {k: {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()} for k, v in kernels.items()},
            'kernels_note': 'one untimed step with every MFMA launch between HIP events; the entry of roofline.kernel is from the timed steps',
        }
        out['mfma_time_share'] = round(sum(v['ms_per_step'] for k, v in kernels.items() if v.get('tflops')) / (1e3 * dt / args.steps), 4) \
            if kernels else None
        # Two more numbers of a one-GPU run, outside the timed region above: what a user of the reference's unchanged
        # loop gets, and BASELINE config 4.  A failure here is reported in place and leaves the headline line intact.
        if world == 1 and not args.no_extras and not args.force_reducer:
            try:
                out['reference_loop'] = reference_loop_rate(model, x, labels)
            except Exception as e:          # noqa: BLE001
                out['reference_loop'] = {'error': '%s: %s' % (type(e).__name__, e)}
            model._engines.clear()
            eng = reducer = None
            progs = []
            torch.cuda.empty_cache()
            try:
                out['config4_densenet201_max_batch'] = config4_rate(device, H, W)
            except Exception as e:          # noqa: BLE001
                out['config4_densenet201_max_batch'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(cfg)
        else:
            out['cpu_baseline'] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if world > 1 or args.force_reducer:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
