#!/usr/bin/env python3
"""Headline benchmark: denoise-step latent-frames/sec on 32x16x16 latent clips (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one VqVideoDiffusionModel.forward over a batch of synthetic random-token clips (config 4:
B = 8 clips per GPU, 32x16x16 latents, codebook 1024, dim 256, dim_head 128, extents 3,3,3, depth 4, mlp 256),
bf16 operands / fp32 accumulation, inputs resident in HBM.  Clips are independent: every rank runs its own
batch, no data-path collective ("weak" scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(B=8, S=32, H=16, W=16, C=1024, dim=256, dim_head=128, heads=1, extents=(3, 3, 3), depth=4, mlp_dim=256)
HBM_PEAK_GBS = 8000.0          # MI355X spec (6290 measured float4 copy), MI355X_MICROARCH.md


def algorithmic_bytes(cfg, elt=2):
    """per layer (5 N D + 8 N I) elt + weights; + tokens + embed write + last-frame logits."""
    N = cfg['B'] * cfg['S'] * cfg['H'] * cfg['W']
    D, I, M, L = cfg['dim'], cfg['dim_head'] * cfg['heads'], cfg['mlp_dim'], cfg['depth']
    layer = (5 * N * D + 8 * N * I) * elt + (4 * D * I + 2 * D * M) * elt
    step = L * layer + N * 8 + N * D * elt + cfg['B'] * cfg['H'] * cfg['W'] * (D * elt + cfg['C'] * 4)
    attn = 4 * N * I * elt
    return step, attn


def ae_layer_model(downscale=2, E=64, P=128, res=64, in_ch=3, elt=2):
    """Per-FRAME flops and tensor traffic of the conv encoder / decoder of the frozen VQ auto-encoder (autoencoder.py:60-152;
    `downscale` Residual pairs, E latent / P hidden planes, res x res frames), layer by layer.  Flops: 2 M K N per convolution
    with the real channel counts.  Bytes: training-mode BatchNorm (quirk Q3) needs a layer's whole-batch statistics before its
    consumer may run, so every normalised tensor makes one round trip per consumer -- the traffic of the launch plan as it
    stands (conv -> [stats] -> conv with prologue -> [stats] -> skip add), bf16 tensors, NOT a lower bound of the problem
    (whose only compulsory traffic is 48 KB of frame in and 2 KB of tokens out).  Returns dicts for encoder and decoder."""
    enc_f = enc_b = 0.0
    px = res * res
    enc_f += 2.0 * px * 9 * in_ch * E
    enc_b += px * in_ch * 4 + px * 8 * elt * 2 + px * E * elt                       # frame in, NHWC8 copy out + in, conv_1 out
    r = res
    for _ in range(downscale):
        for stride in (1, 2):
            pi, po = r * r, (r // stride) ** 2
            enc_f += 2.0 * po * 9 * E * P + 2.0 * po * P * E                        # conv3x3, conv1x1
            enc_b += pi * E * elt + po * P * elt                                    # conv3x3: in, hidden out
            enc_b += po * P * elt + po * E * elt                                    # conv1x1 (BatchNorm + LeakyReLU prologue): in, out
            if stride == 2:
                enc_f += 2.0 * po * 4 * E * E
                enc_b += pi * E * elt + po * E * elt                                # 2x2 / stride 2 skip: in, out
                enc_b += 3 * po * E * elt                                           # BN + BN + add + LeakyReLU: two in, one out
            else:
                enc_b += 3 * po * E * elt                                           # BN + skip + LeakyReLU: two in, one out
            r //= stride
    lat = r * r
    enc_b += lat * E * elt + lat * E * 4 + lat * E * 4 + lat * 8                    # latents -> fp32 -> codebook search -> tokens
    dec_f = dec_b = 0.0
    dec_f += 2.0 * lat * 9 * E * E
    dec_b += 2 * lat * E * elt
    cin = E
    for _ in range(downscale):
        pi, po = r * r, 4 * r * r
        dec_f += 2.0 * po * 9 * cin * P + 2.0 * po * 9 * P * P + 2.0 * po * cin * P
        dec_b += (2 * pi * cin + 2 * po * cin) * elt                                # BN + act; two bilinear x2
        dec_b += (po * cin + po * P) * elt + 2 * po * P * elt                       # conv1; BN + act
        dec_b += (po * cin + po * P) * elt                                          # conv_residual
        dec_b += 3 * po * P * elt                                                   # conv2 + skip
        cin, r = P, 2 * r
    dec_f += 2.0 * r * r * 9 * P * in_ch
    dec_b += r * r * P * elt + r * r * 8 * elt
    return {'enc_flops': enc_f, 'enc_bytes': enc_b, 'dec_flops': dec_f, 'dec_bytes': dec_b,
            'conv1_flops': 2.0 * px * 9 * in_ch * E}


def usable_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota (the GPU box gives 16 of the
    host's cores to one GPU; os.cpu_count() reports the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith('cpu.max'):
                if parts[0] != 'max':
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f2:
                        n = min(n, max(1, q // int(f2.read().split()[0])))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get('WMZ_CPU_BASELINE_THREADS', '16'))))


def log(msg):
    sys.stderr.write(f'[bench {time.strftime("%H:%M:%S")}] {msg}\n')
    sys.stderr.flush()


def encoder_token_agreement(dev, n_frames=16, C=1024):
    """VqAutoEncoder.encode of `n_frames` 64x64 frames on the benched route (compute dtype of the run, BatchNorm in training mode:
    main.py:229-237) against oracle.autoencoder.vqae_encode(training=True) on the same weights: the share of equal tokens, the
    latents' relative error, and whether every differing token is explained by its latent's measured error (the arg-min itself is
    bit-exact on equal inputs; for a differing token the triangle inequality demands
    |x - c_picked| - |x - c_best| <= 2 |x' - x| with x the oracle's latent, x' the route's: tests/test_conv_bf16_oracle_gpu.py)."""
    import torch
    from oracle import autoencoder as oae
    from oracle import vq as ovq
    from world_modelz_amd.train_vqae import VqAutoEncoder
    torch.manual_seed(11)
    ae = VqAutoEncoder(embedding_dim=64, num_embeddings=C, downscale_steps=2, hidden_planes=128).to(dev)
    ae.train()
    sd = {k: v.detach().cpu().clone() for k, v in ae.state_dict().items()}
    frames = torch.rand(n_frames, 3, 64, 64)
    lat_ref = oae.encoder_forward(sd, frames, training=True).permute(0, 2, 3, 1).reshape(-1, 64)
    cb = sd['vq.embedding']
    tok_ref = ovq.encode(lat_ref, cb).reshape(-1)
    with torch.no_grad():
        lat = ae._latents(frames.to(dev))
        tok = ae.vq.encode(lat).reshape(-1).cpu()
    lat = lat.float().cpu().reshape(-1, 64)
    exact_on_own_inputs = bool(torch.equal(tok, ovq.encode(lat, cb).reshape(-1)))
    bad = (tok != tok_ref).nonzero().reshape(-1)
    explained = True
    if bad.numel():
        x, xp = lat_ref[bad].double(), lat[bad].double()
        slack = ((x - cb[0][tok[bad]].double()).norm(dim=-1) - (x - cb[0][tok_ref[bad]].double()).norm(dim=-1)
                 - 2 * (xp - x).norm(dim=-1))
        explained = bool(float(slack.max()) <= 1e-6)
    return {'agreement': 1.0 - bad.numel() / tok.numel(), 'tokens': int(tok.numel()), 'differing': int(bad.numel()),
            'latent_rel_err': float((lat - lat_ref).norm() / lat_ref.norm()),
            'argmin_bit_exact_on_the_routes_own_latents': exact_on_own_inputs,
            'every_difference_within_2x_latent_error': explained,
            'sample': f'{n_frames} frames of 64x64, codebook {C}, random-init weights: a random codebook has many near-ties'}


def cpu_baseline(cfg, sd, budget_s=20.0):
    """The oracle (CPU restatement, kind 'port') on a bounded sample: single clips of the same shape."""
    from oracle import denoiser as oden
    torch.manual_seed(1234)
    z = torch.randint(0, cfg['C'] + 1, (1, cfg['S'], cfg['H'], cfg['W']))
    cores = usable_cores()
    torch.set_num_threads(cores)
    with torch.no_grad():
        oden.denoiser_forward(sd, z, cfg['extents'], cfg['heads'])      # warm-up
        times = []
        t_end = time.time() + budget_s
        while len(times) < 5 and (time.time() < t_end or not times):
            t0 = time.time()
            oden.denoiser_forward(sd, z, cfg['extents'], cfg['heads'])
            times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {'value': cfg['S'] / med, 'unit': 'latent-frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{len(times)} x one 1x{cfg["S"]}x{cfg["H"]}x{cfg["W"]} clip forward (fp32 torch CPU oracle), '
                      f'median {med * 1e3:.0f} ms',
            # the oracle walks the window offsets instead of materialising the unfold, so it is ~9x FASTER than the reference's
            # own CPU path; the reference cannot travel to this box, its number from the build container is quoted beside it
            'reference_elsewhere': {'value': 18.3, 'unit': 'latent-frames/s', 'cores': 8, 'kind': 'reference',
                                    'sample': 'BASELINE.md section 2: the reference\'s own VqVideoDiffusionModel.forward on one '
                                              '1x32x16x16 clip, 1.75 s, torch 2.10 fp32, survey container (8 cores) -- not this host'}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cone', action='store_true', help='skip the secondary last-frame-cone figure (cleaner kernel traces)')
    ap.add_argument('--eager', action='store_true', help='replay the Python launch path instead of the hipGraph')
    ap.add_argument('--train-steps', type=int, default=30, help='timed full training steps reported as train_step (0 = skip)')
    ap.add_argument('--secondary-timeout', type=int, default=420,
                    help='n_gpus > 1: seconds after which a hung secondary figure is abandoned and the headline line printed (0 = never)')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes of torch.distributed.run, BEFORE
        # anything in this process has touched the GPU (torch.cuda.device_count() does not initialise it); rank 0 of the children
        # prints the JSON line on the stdout they inherit.  Fewer visible cards than ranks = a rehearsal: the ranks share cards
        # and the collectives run over gloo (RCCL refuses two ranks on one device); the JSON line says so.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
        sock.close()
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if torch.cuda.device_count() < a.gpus:
            env.setdefault('WMZ_DIST_BACKEND', 'gloo')
            log(f'{torch.cuda.device_count()} device(s) visible for {a.gpus} ranks: rehearsal over {env["WMZ_DIST_BACKEND"]}, ranks share cards')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log('self-launch: ' + ' '.join(cmd))
        sys.exit(subprocess.run(cmd, env=env).returncode)

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks')
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)      # one rank per GPU on a full node; rehearsals may stack ranks on one card
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('WMZ_DIST_BACKEND', 'nccl')       # 'nccl' IS RCCL on ROCm; gloo only for rehearsals
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        # one real collective before anything is timed: the communicator exists, and the rank count the JSON line reports
        # (`rccl_ranks`) is what the collective saw, not what the environment said
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        ranks_seen = int(probe.item())
        assert ranks_seen == world, (ranks_seen, world)
    else:
        backend, ranks_seen = None, 1

    from world_modelz_amd import config, ops
    from world_modelz_amd.main import VqVideoDiffusionModel

    cfg = CFG
    dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    config.set_compute_dtype(dtype)
    torch.manual_seed(42)
    model = VqVideoDiffusionModel(data_shape=(cfg['S'], cfg['H'], cfg['W']), dim=cfg['dim'], num_classes=cfg['C'],
                                  extents=cfg['extents'], depth=cfg['depth'], dim_head=cfg['dim_head'],
                                  mlp_dim=cfg['mlp_dim'], heads=cfg['heads'])
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    gen = torch.Generator().manual_seed(1234 + rank)
    z = torch.randint(0, cfg['C'] + 1, (cfg['B'], cfg['S'], cfg['H'], cfg['W']), generator=gen).to(dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The headline step computes the FULL grid (every plane of every layer), the same work the reference's forward does.
    # The last-frame dependence cone (config.last_frame_cone, the library default for inference) is timed separately below.
    from world_modelz_amd import config as wcfg
    wcfg.set_last_frame_cone(False)
    # one step = one hipGraph launch of the captured forward; --eager replays the Python path
    if a.eager:
        step = lambda: model(z)  # noqa: E731
    else:
        from world_modelz_amd.graph import GraphedForward
        runner = GraphedForward(model, z)
        # the step's input lives in the runner's static token buffer (a producer -- the data loader's device copy, the
        # sampler loop -- writes its tokens there): resident in HBM before the timed region, no staging copy per step
        runner.static_in.copy_(z)
        step = lambda: runner(runner.static_in)  # noqa: E731

    log(f'model built ({"eager" if a.eager else "hipGraph"}), starting {a.warmup} warm-up steps')
    with torch.no_grad():
        # a generational collection of the interpreter (tens of ms once the model, its state_dict copy and the graph
        # runner are alive) must not land between two enqueues of the timed region: collect now, hold the collector off.
        # BEFORE the warm-up: the device idles while the collector runs, and the first ~50 replays after an idle period
        # run 5-10 % slow (clock / power ramp) -- nothing but the barrier may sit between warm-up and timed region
        import gc
        gc.collect()
        gc.disable()
        # untimed pre-warm, whatever W is
        for _ in range(100):
            step()
        for _ in range(a.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        marks = []
        for _ in range(a.steps):
            y = step()
            marks.append(time.perf_counter())
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        try:                                               # context for box-to-box spread: the shader clock right after the timed
            sclk_mhz = int(torch.cuda.clock_rate())        # region (idle: ~100 MHz; the device ramps up under load)
        except Exception:
            sclk_mhz = None
        # (the collector stays off for the secondary figures too; gc.collect() between the sections, in front of each one's
        #  own warm-up, keeps the heap bounded)
        log('warm-up and timed region done')
        if os.environ.get('WMZ_BENCH_MARKS'):
            log('enqueue times per step (ms): ' + ' '.join(f'{(m - t0) * 1e3:.2f}' for m in marks) + f' | synced {(t1 - t0) * 1e3:.2f}')
    elapsed = t1 - t0
    log(f'{a.steps} timed steps in {elapsed:.3f} s')
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(y).all()

    # Per-launch durations of the two kernels that carry the step (the fused per-token layer kernel and the local-3D
    # attention forward), measured live with HIP events on the launch stream (torch's current stream);
    # rocprofv3's averages for the same kernels are in profiles/.
    from world_modelz_amd import fused
    I = cfg['dim_head'] * cfg['heads']
    D = cfg['dim']
    qkv = torch.randn(cfg['B'], cfg['S'], cfg['H'], cfg['W'], 3 * I, device=dev).to(dtype)
    qa, ka, va = qkv[..., :I], qkv[..., I:2 * I], qkv[..., 2 * I:]
    xa = torch.randn(cfg['B'], cfg['S'], cfg['H'], cfg['W'], D, device=dev).to(dtype)
    oa = torch.randn(cfg['B'], cfg['S'], cfg['H'], cfg['W'], I, device=dev).to(dtype)
    layers = list(model.transformer.layers)
    use_fused = dtype == torch.bfloat16 and fused.supported(model.transformer, dtype)

    def run_attn():
        ops.local3d_attention_fwd(qa, ka, va, cfg['extents'], cfg['heads'])

    def run_fused():
        fused.layer_fused(oa, xa, layers[0], layers[1], xflags=3)      # the tiled stream layout, as inside the step

    def time_kernel(fn, reps):
        """Average duration of `reps` back-to-back launches: the launches are captured into one hipGraph (so no host
        gap can sit between them) and the replay is bracketed by one HIP-event pair on the launch stream."""
        side = config.shared_stream('warmup')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        from world_modelz_amd.graph import capture_mode
        with torch.cuda.graph(g, capture_error_mode=capture_mode()):
            for _ in range(reps):
                fn()
        for _ in range(3):                 # clocks / caches settled before the timed replays
            g.replay()
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / reps)
        return sorted(times)[2]

    reps = 50
    with torch.no_grad():
        attn_ms = time_kernel(run_attn, reps)
        fused_ms = time_kernel(run_fused, reps) if use_fused else None

    frames = cfg['B'] * cfg['S'] * world * a.steps
    ms_per_step = elapsed / a.steps * 1e3
    elt = 2 if dtype == torch.bfloat16 else 4
    step_bytes, attn_bytes = algorithmic_bytes(cfg, elt)
    N = cfg['B'] * cfg['S'] * cfg['H'] * cfg['W']
    attn_gbs = attn_bytes / (attn_ms * 1e-3) / 1e9
    attn_roof = {'bound': 'hbm', 'kernel': 'attn_fwd_row16_kernel (local 3D attention forward)', 'achieved': attn_gbs,
                 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': attn_gbs / HBM_PEAK_GBS, 'traffic': None,
                 'algorithmic_bytes_per_launch': attn_bytes, 'avg_launch_ms': attn_ms, 'launches_per_step': cfg['depth'],
                 'launches_timed': reps}
    if fused_ms is not None:
        # per launch: read o [N,I] + x [N,D], write x [N,D] + q [N,I] + k|v [N,2I]  (SURVEY 8d stages K3+K4+K1 of two layers)
        fused_bytes = N * (2 * D + 4 * I) * elt
        fused_gbs = fused_bytes / (fused_ms * 1e-3) / 1e9
        fused_roof = {'bound': 'hbm', 'kernel': 'layer_fused_kernel<head,tail> (to_out+res -> LN -> FF -> res -> next q|k|v)',
                      'achieved': fused_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': fused_gbs / HBM_PEAK_GBS,
                      'traffic': None, 'algorithmic_bytes_per_launch': fused_bytes, 'avg_launch_ms': fused_ms,
                      'launches_per_step': cfg['depth'] - 1, 'launches_timed': reps}
    else:
        fused_roof = None
    # `traffic`: fabric-side bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,
    # gfx950 correction applied: tools/pmc_traffic.py) -- they cannot be collected inside this process, so the committed
    # measurement of the same kernels at the same shapes is reported, but ONLY while the kernel sources still hash to what was
    # measured; otherwise traffic stays null (and `traffic_stale` says why) instead of going stale silently.
    try:
        import hashlib

        def src_hash(names):
            h = hashlib.sha256()
            for n in names:
                with open(os.path.join(ROOT, 'world_modelz_amd', 'csrc', n), 'rb') as fh:
                    h.update(fh.read())
            return h.hexdigest()[:16]

        def newest(fname):
            """(path relative to the repository, parsed JSON) of the newest profiles/rNN/<fname>, or (None, {})."""
            for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, 'profiles')) if d.startswith('r')), reverse=True):
                pth = os.path.join(ROOT, 'profiles', rnd, fname)
                if os.path.exists(pth):
                    with open(pth) as f:
                        return f'profiles/{rnd}/{fname}', json.load(f)
            return None, {}
        tpath, pmc = newest('pmc_traffic.json')
        ipath, pmi = newest('pmc_issue.json')
        for roof, key in ((attn_roof, 'attn_fwd_row16_kernel'), (fused_roof, 'layer_fused_kernel<head,tail>')):
            if roof is None:
                continue
            ent = pmc.get(key)
            if ent is not None:
                if src_hash(ent['sources']) == ent['source_sha16']:
                    roof['traffic'] = ent['traffic_bytes_per_launch']
                    roof['traffic_source'] = f'{tpath} (rocprofv3 PMC passes, same kernel source and shapes)'
                else:
                    roof['traffic_stale'] = f'kernel source changed since {tpath} was measured'
            # MFMA-busy share of the launch (SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), tools/pmc_issue.py), same rule
            ent = pmi.get(key)
            if ent is not None and 'mfma_busy' in ent:
                if src_hash(ent['sources']) == ent['source_sha16']:
                    roof['mfma_busy'] = ent['mfma_busy']
                    roof['issue_counters'] = {k2: ent[k2] for k2 in ('wait_inst_any_share_of_wave_cycles', 'wait_any_share_of_wave_cycles',
                                                                      'active_inst_any_share_of_wave_cycles') if k2 in ent}
                    roof['mfma_busy_source'] = ipath
                else:
                    roof['mfma_busy_stale'] = f'kernel source changed since {ipath} was measured'
    except (OSError, KeyError, ValueError):
        pass
    # `roofline` = the kernel with the larger share of the step.  The per-token kernel runs depth + 1 times per step: depth - 1
    # head + tail launches (the one timed here), the embedding + tail launch and the last layer's head-only launch, which take
    # ~0.6 and ~0.7 of a head + tail launch (profiles/r03/headline_by_grid.csv: 30 and 36 us beside 48-52); attention: depth times.
    dominant_fused = fused_roof is not None and fused_ms * (cfg['depth'] - 1 + 1.3) > attn_ms * cfg['depth']
    out = {
        'metric': 'denoise-step latent-frames/sec (forward, 32x16x16 latent clips)',
        'value': frames / elapsed, 'unit': 'latent-frames/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': a.dtype, 'data': 'synthetic random tokens, random-init weights (seed 42)',
        'config': {'workload': 'BASELINE.json configs[3] per GPU: VqVideoDiffusionModel.forward, B=8 clips/GPU of '
                               '32x16x16 latents, codebook 1024, dim 256, dim_head 128, heads 1, extents 3,3,3, '
                               'depth 4, mlp 256',
                   'clips_per_gpu': cfg['B'], 'latent_shape': [cfg['S'], cfg['H'], cfg['W']], 'codebook': cfg['C'],
                   'parallelism': f'clips sharded over {world} rank(s), no data-path collective'},
        'rccl_ranks': ranks_seen if backend in (None, 'nccl') else 0,
        'dist_backend': {None: None, 'nccl': 'nccl (RCCL)'}.get(backend, f'{backend} (REHEARSAL: ranks share cards, not an RCCL/xGMI figure)'),
        'devices_visible': ndev,
        'roofline': fused_roof if dominant_fused else attn_roof,
        'roofline_other': attn_roof if dominant_fused else fused_roof,
        'launch_mode': 'eager' if a.eager else 'hipGraph replay (1 graph = 1 forward step)',
        'clip_streams': wcfg.get_clip_streams(),
        'sclk_mhz_after_timed_region': sclk_mhz,
        'step_roofline': {'algorithmic_bytes_per_step': step_bytes,
                          'achieved_GBs': step_bytes / (ms_per_step * 1e-3) / 1e9,
                          'frac_of_8TBs': step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          # both denominators -- the 8.0 TB/s specification and the 6.29 TB/s a float4 copy measures
                          'frac_of_measured_copy_6.29TBs': step_bytes / (ms_per_step * 1e-3) / 1e9 / 6290.0},
    }
    gc.collect()
    # The secondary figures below must never cost the headline line: whatever one of them raises (every rank runs the same
    # code on the same shapes, so a failure is the same on all ranks and no rank is left waiting in a collective) is recorded
    # in `secondary_error` and the JSON line is printed with what was measured up to there.  With more than one rank a
    # secondary figure can also HANG (a rank waiting in a collective the others never reach): a watchdog then prints the
    # headline line as measured and ends the rank, so that the launcher sees N clean exits and the line is not lost.
    import threading
    printed = threading.Lock()
    # what the watchdog prints is a COPY of the line taken before the secondary figures start: the main thread goes on
    # adding keys to `out`, and a dict that changes under json.dumps raises
    import copy
    headline_snapshot = copy.deepcopy(out)
    WATCHDOG_EXIT = 3

    def secondary_watchdog():
        """A secondary figure hung (a rank waiting in a collective the others never reach): print the headline as measured and
        end THIS process with a non-zero status -- the launcher and CI must see the hang.  Every rank runs the same timer and it
        stays armed through the final barrier, so the healthy ranks (blocked in that barrier against the one that hung) end the
        same way instead of waiting for the collective's own timeout.  Nothing is restarted or re-executed."""
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            line = headline_snapshot
            line['secondary_error'] = (f'secondary figures did not finish within {a.secondary_timeout} s on {world} ranks '
                                       '(a collective hung?); the headline above was measured before them; exit status '
                                       f'{WATCHDOG_EXIT}')
            line.setdefault('cpu_baseline', None)
            print(json.dumps(line), flush=True)
        sys.stderr.write(f'[bench] rank {rank}: secondary-figure watchdog fired after {a.secondary_timeout} s\n')
        sys.stderr.flush()
        os._exit(WATCHDOG_EXIT)

    watchdog = None
    if world > 1 and a.secondary_timeout > 0:
        watchdog = threading.Timer(a.secondary_timeout, secondary_watchdog)
        watchdog.daemon = True
        watchdog.start()

    def capture_agreed(trainer, example):
        """enable_graph on every rank; None when ALL ranks captured, else the reason (and every rank drops its graph, so that
        no rank replays collectives the others launch eagerly)."""
        err = None
        try:
            trainer.enable_graph(example)
        except Exception as e:  # noqa: BLE001  (a rank whose capture failed must not leave the others in a collective)
            err = f'{type(e).__name__}: {e}'[:300]
        if world > 1:
            ok = torch.tensor([0.0 if err else 1.0], device=dev)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if float(ok.item()) == 0.0:
                err = err or 'capture failed on another rank'
        if err:
            trainer._graph = None
        return err

    # The single-GPU secondary figures (cone, other window, published widths, sampler, frame encoder, VQ-AE step, VQ search, config 2,
    # reference geometry, config 3) are the N = 1 line's: at N > 1 the ranks go straight from the headline to the data-parallel
    # training steps (every rank used to run all of them).
    single_figs = not a.no_cone and world == 1
    try:
        # ---- secondary figure: the same forward with the dead planes elided (bit-identical logits; NOT the headline)
        cone = None
        if use_fused and not a.eager and single_figs:
            wcfg.set_last_frame_cone(True)
            with torch.no_grad():
                crun = GraphedForward(model, z)
                crun.static_in.copy_(z)
                cz = crun.static_in
                yc = crun(cz)
                same = bool(torch.equal(yc, runner(runner.static_in)))
                for _ in range(a.warmup):
                    crun(cz)
                barrier()
                c0 = time.perf_counter()
                for _ in range(a.steps):
                    crun(cz)
                torch.cuda.synchronize()
                barrier()
                cel = time.perf_counter() - c0
            if world > 1:
                t = torch.tensor([cel], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                cel = float(t.item())
            need, src = fused.cone_planes(cfg['S'], cfg['extents'][0], cfg['depth'])
            cone = {'value': frames / cel, 'unit': 'latent-frames/s', 'ms_per_step': cel / a.steps * 1e3,
                    'bit_identical_to_full_grid': same, 'query_planes_per_layer': need, 'source_planes_per_layer': src,
                    'what': 'same logits from the last frame\'s dependence cone only (library default for inference); '
                            'latent-frames counted as for the headline (B*S per step)'}
            wcfg.set_last_frame_cone(False)
            log(f'last-frame cone: {cone["ms_per_step"]:.3f} ms/step, identical={same}')
        out['last_frame_cone'] = cone
        gc.collect()
        # ---- secondary figure: the headline step in the REFERENCE's own precision (all reference arithmetic is fp32: SURVEY 8;
        # main.py:33-36) -- the same model and clips on the library's fp32 mode (op-by-op kernels, exact-f32 MFMA), the step
        # roofline on elt = 4 bytes, and the logits of one clip against the CPU oracle measured in the same run
        fp32m = None
        if single_figs and dtype == torch.bfloat16 and not a.eager:
            with torch.no_grad():
                y16_1 = runner(runner.static_in)[:1].float().cpu()          # (before the dtype moves: the runner re-captures on that)
            config.set_compute_dtype(torch.float32)
            try:
                with torch.no_grad():
                    frun = GraphedForward(model, z)
                    fz = frun.static_in
                    for _ in range(5):
                        y32 = frun(fz)
                    torch.cuda.synchronize()
                    f0_ = time.perf_counter()
                    nf = 10
                    for _ in range(nf):
                        y32 = frun(fz)
                    torch.cuda.synchronize()
                    fel32 = (time.perf_counter() - f0_) / nf
                    from oracle import denoiser as _oden
                    torch.set_num_threads(usable_cores())
                    ref1 = _oden.denoiser_forward(sd_cpu, z[:1].cpu(), cfg['extents'], cfg['heads'])
                    rel32 = float((y32[:1].float().cpu() - ref1).norm() / ref1.norm())
                    rel16 = float((y16_1 - ref1).norm() / ref1.norm())
                b32, _ = algorithmic_bytes(cfg, elt=4)
                step_flops = 184.5e9                        # algorithmic flops per forward step at config 4
                fp32m = {'ms_per_step': fel32 * 1e3, 'value': cfg['B'] * cfg['S'] / fel32, 'unit': 'latent-frames/s', 'dtype': 'f32',
                         # at 1/16 of the matrix rate this step is COMPUTE-bound (floor 184.5 GF / 157 TF/s = 1.17 ms): the matrix
                         # roof is the one that binds; the HBM figure on elt = 4 bytes beside it
                         'roofline': {'bound': 'mfma-f32', 'achieved': step_flops / fel32 / 1e12, 'peak': 157.0, 'unit': 'TFLOP/s',
                                      'frac': step_flops / fel32 / 1e12 / 157.0, 'algorithmic_flops_per_step': step_flops},
                         'roofline_hbm': {'bound': 'hbm', 'achieved': b32 / fel32 / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                          'frac': b32 / fel32 / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_step': b32},
                         'logits_rel_err_vs_cpu_oracle_one_clip': rel32, 'bf16_logits_rel_err_same_clip': rel16,
                         'what': 'the headline forward step with fp32 activations and fp32 MFMA arithmetic (the reference\'s precision), full grid, one hipGraph'}
                log(f'fp32 mode: {fel32 * 1e3:.3f} ms/step, logits rel err vs oracle {rel32:.2e} (bf16: {rel16:.2e})')
                del frun
            finally:
                config.set_compute_dtype(dtype)
        out['fp32_mode'] = fp32m
        gc.collect()
        # ---- secondary figure: the PRECISE fused mode (config.compute_dtype(torch.float16)): the headline step on the same fused
        # kernels with IEEE-half MFMA operands and a half stream -- north_star's "within 1e-3 rel on attention logits" end to end,
        # at the matrix rate of bf16.  Logits of one clip against the CPU oracle measured in the same run.
        prec = None
        if single_figs and dtype == torch.bfloat16 and not a.eager and fp32m is not None:
            config.set_compute_dtype(torch.float16)
            try:
                with torch.no_grad():
                    prun = GraphedForward(model, z)
                    pz = prun.static_in
                    for _ in range(50):
                        yp = prun(pz)
                    torch.cuda.synchronize()
                    # (best of three groups of 50: this figure is taken right behind the fp32 mode's 4-ms steps, and the first
                    #  group still runs at the clocks those left behind -- same-process A/B against bf16: +3 %, not +10 %)
                    npz_ = 50
                    felp = float('inf')
                    for _ in range(3):
                        p0_ = time.perf_counter()
                        for _ in range(npz_):
                            yp = prun(pz)
                        torch.cuda.synchronize()
                        felp = min(felp, (time.perf_counter() - p0_) / npz_)
                    relp = float((yp[:1].float().cpu() - ref1).norm() / ref1.norm())
                bp, _ = algorithmic_bytes(cfg, elt=2)
                prec = {'ms_per_step': felp * 1e3, 'value': cfg['B'] * cfg['S'] / felp, 'unit': 'latent-frames/s', 'dtype': 'f16',
                        'roofline': {'bound': 'hbm', 'achieved': bp / felp / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                     'frac': bp / felp / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_step': bp},
                        'roofline_mfma': {'bound': 'mfma', 'achieved': 184.5e9 / felp / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s',
                                          'frac': 184.5e9 / felp / 1e12 / 2500.0,
                                          'what': 'the f16 matrix pipe (same dense peak as bf16), the roof this mode computes on'},
                        'logits_rel_err_vs_cpu_oracle_one_clip': relp, 'bf16_logits_rel_err_same_clip': rel16,
                        'fp32_logits_rel_err_same_clip': rel32, 'gate': 'logits_rel_err <= 1e-3 and ms_per_step <= 1.5',
                        'what': 'the headline forward step on the fused kernels with IEEE-half operands and stream (fp32 accumulation, '
                                'fp32 LayerNorm / softmax / GELU arithmetic, fp32 last-frame projection), full grid, one hipGraph'}
                log(f'precise (f16) mode: {felp * 1e3:.3f} ms/step, logits rel err vs oracle {relp:.2e}')
                del prun
            finally:
                config.set_compute_dtype(dtype)
        out['precise_mode'] = prec
        gc.collect()
        # ---- secondary figure: the same step with the reference's other published attention window, 7 x 3 x 3 (extents 3, 1, 1:
        # BASELINE.md run-03), full grid, same model otherwise
        win = None
        if use_fused and not a.eager and single_figs:
            torch.manual_seed(42)
            m2 = VqVideoDiffusionModel(data_shape=(cfg['S'], cfg['H'], cfg['W']), dim=cfg['dim'], num_classes=cfg['C'],
                                       extents=(3, 1, 1), depth=cfg['depth'], dim_head=cfg['dim_head'], mlp_dim=cfg['mlp_dim'],
                                       heads=cfg['heads']).to(dev).eval()
            with torch.no_grad():
                wrun = GraphedForward(m2, z)
                wz = wrun.static_in
                for _ in range(50 + a.warmup):
                    wrun(wz)
                barrier()
                w0 = time.perf_counter()
                for _ in range(a.steps):
                    wrun(wz)
                torch.cuda.synchronize()
                barrier()
                wel = time.perf_counter() - w0
            if world > 1:
                t = torch.tensor([wel], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                wel = float(t.item())
            win = {'value': frames / wel, 'unit': 'latent-frames/s', 'ms_per_step': wel / a.steps * 1e3,
                   'what': 'the headline step with the 7x3x3 window of the published run-03 (extents 3,1,1) instead of 7x7x7'}
            log(f'7x3x3 window: {win["ms_per_step"]:.3f} ms/step')
            del wrun, m2
        out['window_7x3x3'] = win
        gc.collect()
        # ---- secondary figure: the reference's two PUBLISHED runs (results/README.md: dim 96 / mlp 256 / depth 12 and dim 384 /
        # mlp 512 / depth 20, one head of 128, window 7x3x3), same clips.  Their widths are outside the default-width per-token
        # kernel (register-chained 32-token waves: DESIGN.md 4.2) and run on csrc/layer_chain.hip (16-token waves, same fusion: one
        # attention launch + one per-token launch per layer); one hipGraph per step.
        pub = None
        if not a.eager and single_figs and dtype == torch.bfloat16:
            pub = []
            for dim_, mlp_, depth_ in ((96, 256, 12), (384, 512, 20)):
                torch.manual_seed(42)
                m3 = VqVideoDiffusionModel(data_shape=(cfg['S'], cfg['H'], cfg['W']), dim=dim_, num_classes=cfg['C'], extents=(3, 1, 1),
                                           depth=depth_, dim_head=128, mlp_dim=mlp_, heads=1).to(dev).eval()
                with torch.no_grad():
                    prun = GraphedForward(m3, z)
                    pz = prun.static_in
                    for _ in range(10):
                        prun(pz)
                    barrier()
                    pel = float('inf')               # (best of three groups of 10: a box hiccup once put 7.99 ms into one group of a 0.86-ms step)
                    for _ in range(3):
                        p0 = time.perf_counter()
                        for _ in range(10):
                            prun(pz)
                        torch.cuda.synchronize()
                        pel = min(pel, (time.perf_counter() - p0) / 10)
                entry = {'dim': dim_, 'mlp_dim': mlp_, 'depth': depth_, 'extents': [3, 1, 1], 'ms_per_step': pel * 1e3,
                         'value': cfg['B'] * cfg['S'] / pel, 'unit': 'latent-frames/s',
                         'params': sum(p.numel() for p in m3.parameters())}
                log(f'published widths dim {dim_} depth {depth_}: {pel * 1e3:.3f} ms/step')
                del prun
                if a.train_steps > 0 and world == 1:
                    # ... and their TRAINING step (the reference's published runs are training runs): op-by-op GEMM path -- no
                    # fused per-token kernels at these widths --, one hipGraph per step
                    from world_modelz_amd.train import DenoiserTrainer as _PT
                    m3.train()
                    ptr = _PT(m3, cfg['C'], lr=1e-4, warmup=500, max_steps=200000, distributed=False)
                    ptr.enable_graph(z)
                    pr = torch.full((cfg['B'],), 0.5)
                    for _ in range(2):
                        ptr.train_step(z, r=pr)
                    torch.cuda.synchronize()
                    p0 = time.perf_counter()
                    for _ in range(5):
                        ptr.train_step(z, r=pr)
                    torch.cuda.synchronize()
                    entry['train_ms_per_step'] = (time.perf_counter() - p0) / 5 * 1e3
                    entry['train_launch_mode'] = 'hipGraph (chain kernels: per layer one attention forward + one per-token launch, attention backward between two per-token backward launches, one weight-gradient launch pair on the side branch)'
                    log(f'published widths dim {dim_} depth {depth_}: training step {entry["train_ms_per_step"]:.2f} ms')
                    del ptr
                pub.append(entry)
                del m3
                gc.collect()
        out['published_run_widths'] = pub
        # ---- secondary figure: the sampler loop (main.py:50-117): one denoise iteration = draw + re-mask + forward
        # on the last frame's dependence cone, ONE hipGraph launch (sample.py); B clips, top-k 100
        smp = None
        if use_fused and not a.eager and single_figs:
            from world_modelz_amd.sample import sample_frames
            wcfg.set_last_frame_cone(True)
            # the first call of a configuration captures the sampler step (kept with the model: sample._Session); timed: the steady state
            sample_frames(model, z.clamp(max=cfg['C'] - 1), cfg['C'], 1, num_eval_iterations=30, sample_topk=100)
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            sample_frames(model, z.clamp(max=cfg['C'] - 1), cfg['C'], 2, num_eval_iterations=30, sample_topk=100)
            torch.cuda.synchronize()
            sel = (time.perf_counter() - s0) / 60
            wcfg.set_last_frame_cone(False)
            smp = {'ms_per_iteration': sel * 1e3, 'value': cfg['B'] / (30 * sel), 'unit': 'generated latent-frames/s (30 iterations each)',
                   'what': f"sample_frames: 2 frames x 30 denoise iterations (29 forwards + 30 draws a frame) for {cfg['B']} clips, top-k 100, "
                           'the captured step reused from a first call'}
            log(f'sampler: {sel * 1e3:.3f} ms per denoise iteration')
        out['sampler_iteration'] = smp
        # ---- secondary figure: the stage in front of the denoiser: the frozen VQ auto-encoder turning frames into
        # latent tokens -- conv encoder (NHWC implicit GEMM, BatchNorm in train mode: quirk Q3) + codebook argmin.  B*S frames
        # of 64x64 RGB -> 16x16 tokens each (2 down-scale steps), codebook 1024 x 64.
        frame_enc = None
        if single_figs:
            from world_modelz_amd.train_vqae import VqAutoEncoder
            torch.manual_seed(7)
            ae = VqAutoEncoder(embedding_dim=64, num_embeddings=cfg['C'], downscale_steps=2, hidden_planes=128).to(dev)
            frames_in = torch.randn(cfg['B'] * cfg['S'], 3, 64, 64, device=dev)
            with torch.no_grad():
                if a.eager:
                    enc = ae.encode
                else:
                    from world_modelz_amd.graph import GraphedEncoder
                    enc = GraphedEncoder(ae, frames_in)
                # (as for the headline: the batch is resident where the step reads it -- the runner's static input -- when the timed
                #  region starts; enc(other_tensor) adds one 12.6 MB device copy in front of the launch)
                fin = frames_in if a.eager else enc.static_in
                if not a.eager:
                    fin.copy_(frames_in)
                for _ in range(2):
                    tok = enc(fin)
                barrier()
                f0 = time.perf_counter()
                for _ in range(10):
                    tok = enc(fin)
                torch.cuda.synchronize()
                fel = (time.perf_counter() - f0) / 10
            assert tok.shape == (cfg['B'] * cfg['S'], 16, 16)
            aem = ae_layer_model()
            nfr = cfg['B'] * cfg['S']
            vq_flops = 3.0 * 256 * cfg['C'] * 64                   # per frame: 256 latents x C codes x E (sub, mul, add)
            frame_enc = {'value': cfg['B'] * cfg['S'] / fel, 'unit': 'frames/s', 'ms_per_batch': fel * 1e3,
                         # HBM: the launch plan's tensor traffic (training-mode BatchNorm forces a round trip per normalised
                         # tensor: ae_layer_model); MFMA: the convolutions' 2 M K N at the dense bf16 peak
                         'roofline': {'bound': 'hbm', 'achieved': aem['enc_bytes'] * nfr / fel / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                      'frac': aem['enc_bytes'] * nfr / fel / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                                      'bytes_per_frame': aem['enc_bytes'], 'what': 'tensor traffic of the launch plan (one round trip per BatchNorm-ed tensor and consumer), bf16'},
                         'roofline_mfma': {'bound': 'mfma', 'achieved': (aem['enc_flops'] + vq_flops) * nfr / fel / 1e12, 'peak': 2500.0,
                                           'unit': 'TFLOP/s', 'frac': (aem['enc_flops'] + vq_flops) * nfr / fel / 1e12 / 2500.0,
                                           'flops_per_frame': aem['enc_flops'] + vq_flops},
                         'launch_mode': 'eager' if a.eager else 'hipGraph (GraphedEncoder: one launch per batch of frames)',
                         'what': f"VqAutoEncoder.encode of {cfg['B'] * cfg['S']} 64x64 RGB frames -> 16x16 tokens "
                                 '(BatchNorm in training mode as in main.py:236, quirk Q3)'}
            # fabric-side bytes of one call, every launch summed (tools/pmc_encode_total.py, committed under profiles/): reported
            # only while the conv / VQ sources still hash to what was measured
            try:
                ppath, pme = newest('pmc_frame_encoder.json')
                if pme:
                    if src_hash(pme['sources']) == pme['source_sha16'] and pme['frames_per_call'] == nfr:
                        frame_enc['roofline']['traffic'] = pme['traffic_bytes_per_call']
                        frame_enc['roofline']['traffic_per_frame'] = pme['traffic_bytes_per_call'] / nfr
                        frame_enc['roofline']['traffic_source'] = f'{ppath} (rocprofv3 PMC passes over tools/prof_encode.py, all launches of one call)'
                    else:
                        frame_enc['roofline']['traffic_stale'] = f'kernel sources (or the batch) changed since {ppath} was measured'
            except (OSError, KeyError, ValueError, NameError):
                pass
            # how far the benched route's tokens are from the reference's: 16 frames through THIS route (bf16 direct kernels,
            # training-mode BatchNorm) against the fp32 CPU oracle on the same weights -- outside the timed loop, oracle as checker
            if rank == 0 and not a.no_cpu_baseline:
                try:
                    frame_enc['token_agreement_vs_fp32_oracle'] = encoder_token_agreement(dev)
                    log(f"frame encoder tokens vs fp32 oracle: {frame_enc['token_agreement_vs_fp32_oracle']}")
                except Exception as e:                                   # noqa: BLE001  (a diagnostic must not cost the bench line)
                    frame_enc['token_agreement_vs_fp32_oracle'] = {'error': f'{type(e).__name__}: {e}'[:300]}
            log(f'frame encoder: {fel * 1e3:.2f} ms per {cfg["B"] * cfg["S"]} frames')
        out['frame_encoder'] = frame_enc
        # ---- secondary figure: one VQ-AE training step (train_vqae.py:125-164: encoder -> VectorQuantizerEMA incl. the EMA
        # codebook update -> decoder, SmoothL1 + commitment loss, backward, AdamW) on 64 frames of 64x64, eager launches
        vqae = None
        if single_figs and world == 1:
            from world_modelz_amd.train import VqaeTrainer
            torch.manual_seed(7)
            ae2 = VqAutoEncoder(embedding_dim=64, num_embeddings=cfg['C'], downscale_steps=2, hidden_planes=128).to(dev)
            vt = VqaeTrainer(ae2, distributed=False)
            fr = torch.rand(64, 3, 64, 64, device=dev)
            if not a.eager:
                vt.enable_graph(fr)
            for _ in range(3):
                vt.train_step(fr)
            torch.cuda.synchronize()
            v0 = time.perf_counter()
            for _ in range(5):
                vt.train_step(fr)
            torch.cuda.synchronize()
            vel = (time.perf_counter() - v0) / 5
            aem = ae_layer_model()
            # forward + data gradient + weight gradient of every convolution (the first layer has no data gradient)
            vq_tr_flops = 64 * (3.0 * (aem['enc_flops'] + aem['dec_flops']) - aem['conv1_flops'])
            vqae = {'value': 64 / vel, 'unit': 'frames/s', 'ms_per_step': vel * 1e3,
                    'roofline': {'bound': 'mfma', 'achieved': vq_tr_flops / vel / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s',
                                 'frac': vq_tr_flops / vel / 1e12 / 2500.0, 'flops_per_step': vq_tr_flops,
                                 'what': '3 x the convolutions 2 M K N (forward, data gradient, weight gradient) of encoder + decoder on 64 frames'},
                    'launch_mode': 'eager' if a.eager else 'hipGraph',
                    'what': 'VqaeTrainer.train_step on 64 RGB frames of 64x64 (codebook 1024 x 64, 2 down-scale steps, 128 planes): one hipGraph replay + one host read-back per step'}
            log(f'VQ-AE training step: {vel * 1e3:.2f} ms per 64 frames')
            del vt, ae2
        out['vqae_train_step'] = vqae
        # ---- secondary figure: the VQ codebook nearest-neighbour micro-bench of SURVEY 8(d): x = randn(N, 64), codebook =
        # randn(C, 64), N = 65 536, C in {512, 1024, 8192}, seed 0.  Two ways to the SAME bits (indices and minimum distances in the
        # reference's fp32 summation order): the screened search (csrc/vq_screen.hip: bf16 head/tail products on the matrix cores
        # with a proven error bound pick each row's code, undecided rows are re-scanned exactly) -- bound: the matrix pipe, 3 bf16
        # MFMA products per (n, c, e) = 6 N C E flops against the dense bf16 peak; and the full scan in the pinned arithmetic
        # (csrc/vq.hip) -- three UN-fused fp32 lane operations per (n, c, e), bound: the vector ALU's lane-op rate (half the
        # 157.3 TFLOP/s FMA peak).  HBM traffic (N*E*4 in, N*8 out, codebook resident) is two orders below its roof.
        vq = None
        if single_figs:
            vq = []
            gvq = torch.Generator(device='cpu').manual_seed(0)
            Nq, Eq = 65536, 64
            xq = torch.randn(Nq, Eq, generator=gvq).to(dev)
            for Cq in (512, 1024, 8192):
                cbq = torch.randn(Cq, Eq, generator=gvq).to(dev)
                vq_ms = time_kernel(lambda: ops.vq_argmin(xq, cbq), 10)
                ex_ms = time_kernel(lambda: ops.vq_argmin(xq, cbq, exact_scan=True), 10)
                lane_ops = 3.0 * Nq * Cq * Eq
                mfma_flops = 6.0 * Nq * Cq * Eq
                vq.append({'N': Nq, 'C': Cq, 'E': Eq, 'ms': vq_ms, 'rows_per_s': Nq / (vq_ms * 1e-3),
                           'roofline': {'bound': 'mfma', 'achieved': mfma_flops / (vq_ms * 1e-3) / 1e12, 'peak': 2500.0,
                                        'unit': 'TFLOP/s', 'frac': mfma_flops / (vq_ms * 1e-3) / 1e12 / 2500.0},
                           'exact_scan': {'ms': ex_ms, 'roofline': {'bound': 'valu-f32', 'achieved': lane_ops / (ex_ms * 1e-3) / 1e12,
                                                                    'peak': 78.65, 'unit': 'T lane-op/s',
                                                                    'frac': lane_ops / (ex_ms * 1e-3) / 1e12 / 78.65}},
                           'hbm_GBs': (Nq * Eq * 4 + Nq * 8) / (vq_ms * 1e-3) / 1e9})
            log('vq argmin: ' + ', '.join(f"C={v['C']} {v['ms'] * 1e3:.0f} us screened / {v['exact_scan']['ms'] * 1e3:.0f} us exact scan" for v in vq))
        out['vq_argmin'] = vq
        gc.collect()
        # ---- secondary figure: BASELINE configs[1] -- ONE Local3dAttention.forward on an 8x8x8 latent grid, d = 256, one head of
        # 128, bf16 (SURVEY 8d inputs: x = LN(randn(1,8,8,8,256)), q = randn(same), seed 0), window 7x7x7 and 7x3x3.  512 tokens:
        # two workgroups' worth of attention and four small GEMMs, so the figure is launch latency, not throughput; one hipGraph
        # of 20 forwards, HIP events around the replay.
        cfg2 = None
        if single_figs and dtype == torch.bfloat16:
            from world_modelz_amd.local_3d_attention import Local3dAttention
            cfg2 = []
            for ext2 in ((3, 3, 3), (3, 1, 1)):
                torch.manual_seed(0)
                att = Local3dAttention(ext2, 256, heads=1, dim_head=128).to(dev).eval()
                x2 = torch.nn.functional.layer_norm(torch.randn(1, 8, 8, 8, 256), (256,)).to(dev)
                q2 = torch.randn(1, 8, 8, 8, 256).to(dev)
                with torch.no_grad():
                    us = time_kernel(lambda: att(x2, q2), 20) * 1e3
                cfg2.append({'extents': list(ext2), 'us_per_forward': us, 'value': 8 / (us * 1e-6), 'unit': 'latent-frames/s',
                             'what': 'Local3dAttention.forward(x, q): to_q / to_k / to_v, local 3D attention core, to_out; fp32 in / out at '
                                     'the module boundary, bf16 inside; B=1, 8x8x8 grid, dim 256, 1 head x 128'})
            log('config 2: ' + ', '.join(f"{c['extents']} {c['us_per_forward']:.1f} us" for c in cfg2))
        out['config2_attention'] = cfg2
        # ---- secondary figure: the reference's OWN geometry and published model (main.py:394 `data_shape=z.shape # (6,8,8)`,
        # train_vqae.py:83-86: 3 down-scale steps -> 8x8 latents of 64x64 frames; results/README.md:13-22: run 03, dim 384 / mlp 512 /
        # 20 layers / one head of 128 / window (7,3,3) = --extent 3,1,1 / batch 64 / 5 context frames + 1): forward denoise step and
        # full training step, one hipGraph each.  8-wide planes take the row attention kernels as tile rows of 16 (round 4).
        refgeo = None
        if not a.eager and single_figs and dtype == torch.bfloat16:
            from world_modelz_amd.train import DenoiserTrainer as _RT
            torch.manual_seed(42)
            mr = VqVideoDiffusionModel(data_shape=(6, 8, 8), dim=384, num_classes=512, extents=(3, 1, 1), depth=20, dim_head=128,
                                       mlp_dim=512, heads=1).to(dev).eval()
            zr = torch.randint(0, 513, (64, 6, 8, 8), generator=gen).to(dev)
            with torch.no_grad():
                wcfg.set_last_frame_cone(False)
                rrun = GraphedForward(mr, zr)
                for _ in range(10):
                    rrun(rrun.static_in)
                torch.cuda.synchronize()
                r0_ = time.perf_counter()
                for _ in range(10):
                    rrun(rrun.static_in)
                torch.cuda.synchronize()
                rf = (time.perf_counter() - r0_) / 10
            refgeo = {'forward_ms_per_step': rf * 1e3, 'forward_value': 64 * 6 / rf, 'unit': 'latent-frames/s',
                      'params': sum(p.numel() for p in mr.parameters()),
                      'what': 'reference geometry: B = 64 clips of (6, 8, 8) latents, codebook 512, dim 384 / mlp 512 / depth 20 / '
                              '1 x 128 / extents (3,1,1) (results/README.md run 03); full grid forward and one training step'}
            del rrun
            # ... the same forward in the precise mode (half unit of the chain kernel + half attention, round 6)
            config.set_compute_dtype(torch.float16)
            try:
                with torch.no_grad():
                    rrun = GraphedForward(mr, zr)
                    for _ in range(10):
                        rrun(rrun.static_in)
                    torch.cuda.synchronize()
                    r0_ = time.perf_counter()
                    for _ in range(10):
                        rrun(rrun.static_in)
                    torch.cuda.synchronize()
                    refgeo['precise_forward_ms_per_step'] = (time.perf_counter() - r0_) / 10 * 1e3
                del rrun
            finally:
                config.set_compute_dtype(dtype)
            if a.train_steps > 0 and world == 1:
                mr.train()
                rt = _RT(mr, 512, lr=1e-4, warmup=500, max_steps=75000, distributed=False)
                rt.enable_graph(zr.clamp(max=511))
                rr = torch.full((64,), 0.5)
                for _ in range(3):
                    rt.train_step(zr.clamp(max=511), r=rr)
                torch.cuda.synchronize()
                r0_ = time.perf_counter()
                for _ in range(5):
                    rt.train_step(zr.clamp(max=511), r=rr)
                torch.cuda.synchronize()
                rtr = (time.perf_counter() - r0_) / 5
                refgeo['train_ms_per_step'] = rtr * 1e3
                refgeo['train_value'] = 64 * 6 / rtr
                del rt
            log(f"reference geometry 64 x (6,8,8), dim 384 / depth 20: forward {refgeo['forward_ms_per_step']:.3f} ms "
                f"(precise mode {refgeo['precise_forward_ms_per_step']:.3f}), "
                f"train {refgeo.get('train_ms_per_step', float('nan')):.2f} ms")
            del mr
            gc.collect()
        out['reference_geometry'] = refgeo
        # ---- secondary figure: BASELINE configs[2] -- the training step of vq-video-diffusion/main.py on B = 16 clips of 16x16x16
        # latents (same token count per step as the headline, shorter clips: more border planes), codebook 1024; one hipGraph per step
        cfg3 = None
        if a.train_steps > 0 and single_figs and world == 1 and not a.eager:
            from world_modelz_amd.train import DenoiserTrainer as _DT
            torch.manual_seed(42)
            m3 = VqVideoDiffusionModel(data_shape=(16, 16, 16), dim=cfg['dim'], num_classes=cfg['C'], extents=cfg['extents'],
                                       depth=cfg['depth'], dim_head=cfg['dim_head'], mlp_dim=cfg['mlp_dim'], heads=cfg['heads']).to(dev)
            z3 = torch.randint(0, cfg['C'], (16, 16, 16, 16), generator=gen).to(dev)
            t3 = _DT(m3, cfg['C'], lr=1e-4, warmup=500, max_steps=200000, distributed=False)
            r3 = torch.full((16,), 0.5)
            t3.enable_graph(z3)
            for _ in range(5):
                t3.train_step(z3, r=r3)
            torch.cuda.synchronize()
            c0_ = time.perf_counter()
            n3 = max(5, a.train_steps // 2)
            for _ in range(n3):
                t3.train_step(z3, r=r3)
            torch.cuda.synchronize()
            c3 = (time.perf_counter() - c0_) / n3
            with torch.no_grad():
                m3.eval()
                wcfg.set_last_frame_cone(False)
                f3run = GraphedForward(m3, z3)
                for _ in range(20):
                    f3run(f3run.static_in)
                torch.cuda.synchronize()
                c0_ = time.perf_counter()
                for _ in range(a.steps):
                    f3run(f3run.static_in)
                torch.cuda.synchronize()
                f3 = (time.perf_counter() - c0_) / a.steps
            # ... and the reference's step END TO END (main.py:229-237: every training step first encodes its B x (n_past + 1)
            # frames with the frozen auto-encoder -- BatchNorm in training mode, quirk Q3 -- then corrupts and trains): 256 fp32
            # frames of 64x64 in -> encode (one hipGraph) -> tokens [16, 16, 16, 16] -> the graphed training step.  One number.
            e2e = None
            if frame_enc is not None and not a.eager:
                fr3 = torch.randn(16 * 16, 3, 64, 64, device=dev)

                def e2e_step():
                    tok3 = enc(fr3)
                    return t3.train_step(tok3.view(16, 16, 16, 16), r=r3)
                for _ in range(3):
                    e2e_step()
                torch.cuda.synchronize()
                c0_ = time.perf_counter()
                for _ in range(n3):
                    e2e_step()
                torch.cuda.synchronize()
                ce = (time.perf_counter() - c0_) / n3
                e2e = {'ms_per_step': ce * 1e3, 'value': 16 * 16 / ce, 'unit': 'latent-frames/s',
                       'encode_ms': frame_enc['ms_per_batch'], 'train_ms': c3 * 1e3,
                       'what': 'main.py:229-287 end to end: 256 fp32 RGB frames of 64x64 -> VqAutoEncoder.encode (BatchNorm in training '
                               'mode, one hipGraph) -> token corruption + forward + CE + backward + AdamW (one hipGraph) on 16 clips of 16x16x16'}
                log(f'config 3 end to end (frames -> tokens -> training step): {ce * 1e3:.2f} ms/step')
            out['config3_end_to_end'] = e2e
            cfg3 = {'train_ms_per_step': c3 * 1e3, 'train_value': 16 * 16 / c3, 'forward_ms_per_step': f3 * 1e3,
                    'forward_value': 16 * 16 / f3, 'unit': 'latent-frames/s',
                    'what': 'B=16 clips of 16x16x16 latents, codebook 1024, default model: full training step (one hipGraph) and the '
                            'forward denoise step (full grid)'}
            log(f'config 3: train {c3 * 1e3:.2f} ms/step, forward {f3 * 1e3:.3f} ms/step')
            del t3, m3, f3run
            gc.collect()
        out['config3_train_step'] = cfg3
        out.setdefault('config3_end_to_end', None)
        # ---- secondary figure: the full training step (corrupt -> forward -> CE -> backward -> grad-norm -> AdamW -> operand
        # re-pack), same shapes, same rules (barrier + sync both sides, max over ranks).  One GPU: the whole step is ONE hipGraph
        # replay (DenoiserTrainer.enable_graph) plus the loss-aware sampler's one host read-back per step.  n_gpus > 1: the same
        # graph with the per-layer gradient all-reduce buckets captured on a side stream (the eager path is timed beside it).
        train = None
        if a.train_steps > 0:
            from world_modelz_amd.train import DenoiserTrainer, corrupt_last_frame
            model.train()
            rfix = torch.full((cfg['B'],), 0.5)
            single_ms = None
            if world > 1 and not a.eager:
                # the SAME step without the data-parallel reducer, in this process, on a copy of the model, before the reducer
                # exists: what one rank does alone on its B clips.  Per-GPU work is fixed (weak scaling), so
                # scaling_efficiency = single-rank step time / data-parallel step time is the fraction of N x the single-GPU
                # throughput the N ranks reach together (north_star: >= 6x at 8 GPUs <=> efficiency >= 0.75).
                import copy as _copy
                m1 = _copy.deepcopy(model)
                t1 = DenoiserTrainer(m1, cfg['C'], lr=1e-4, warmup=500, max_steps=200000, distributed=False)
                t1.enable_graph(z)
                for _ in range(5):
                    t1.train_step(z, r=rfix)
                barrier()
                s0_ = time.perf_counter()
                for _ in range(a.train_steps):
                    t1.train_step(z, r=rfix)
                torch.cuda.synchronize()
                barrier()
                sel_ = time.perf_counter() - s0_
                tt = torch.tensor([sel_], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                single_ms = float(tt.item()) / a.train_steps * 1e3
                log(f'single-rank reference training step (no reducer): {single_ms:.3f} ms')
                del t1, m1
                gc.collect()
            tr = DenoiserTrainer(model, cfg['C'], lr=1e-4, warmup=500, max_steps=200000, distributed=world > 1)
            # hipGraph replay is the launch mode at every world size: the per-layer RCCL all-reduces are captured with the step
            # (side-stream fork / join inside the graph).  gloo (rehearsals on shared cards) cannot be captured: eager there.
            graphed = not a.eager and (world == 1 or backend == 'nccl')

            def eager_step():
                tr.arena.zero_grad()
                zc, tgt = corrupt_last_frame(z, rfix, cfg['C'])
                tr.forward_backward(zc, tgt)
                tr.optimizer_step()

            def timed(fn, n):
                for _ in range(5):
                    fn()
                barrier()
                t_0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                barrier()
                el = time.perf_counter() - t_0
                if world > 1:
                    tt = torch.tensor([el], device=dev, dtype=torch.float64)
                    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                    el = float(tt.item())
                return el

            eager_ms, overlap = None, None
            if world > 1 or not graphed:
                # the eager launch path (host-bound at ~80 launches per step): timed for the record, and the only mode in which
                # HIP events can bracket the collectives on the side stream (grad_allreduce_overlap)
                if tr.reducer is not None:
                    tr.reducer.enable_timing()
                n_e = max(5, a.train_steps // 3)
                eager_ms = timed(eager_step, n_e) / n_e * 1e3
                if tr.reducer is not None:
                    overlap = tr.reducer.timing_summary()
                    tr.reducer.enable_timing(False)
            graph_error = None
            if graphed:
                graph_error = capture_agreed(tr, z)
                if graph_error:
                    log('hipGraph capture of the data-parallel step failed, eager launches instead: ' + graph_error)
                    graphed = False
            if graphed:
                tel = timed(lambda: tr.train_step(z, r=rfix), a.train_steps)
            elif eager_ms is None:
                tel = timed(eager_step, a.train_steps)
            else:
                tel = eager_ms * 1e-3 * a.train_steps
            tms = tel / a.train_steps * 1e3
            # a training step is ~3x the forward's flops (no recompute of the attention core: the backward works
            # from the saved log-sum-exp); bytes likewise ~3x the forward's algorithmic bytes (activations written once, read by
            # the backward, gradients written once)
            D_, I_, M_, L_ = cfg['dim'], cfg['dim_head'] * cfg['heads'], cfg['mlp_dim'], cfg['depth']
            Kw = (2 * cfg['extents'][0] + 1) * (2 * cfg['extents'][1] + 1) * (2 * cfg['extents'][2] + 1)
            fwd_flops = L_ * (6.0 * N * D_ * I_ + 4.0 * N * Kw * I_ + 2.0 * N * I_ * D_ + 4.0 * N * D_ * M_) \
                + 2.0 * cfg['B'] * cfg['H'] * cfg['W'] * D_ * cfg['C']
            train = {'value': cfg['B'] * cfg['S'] * world * a.train_steps / tel, 'unit': 'latent-frames/s',
                     'ms_per_step': tms, 'steps': a.train_steps,
                     'what': 'corrupt + forward + CE + backward + grad-norm + AdamW + operand re-pack: '
                             + ('ONE hipGraph replay per step + the sampler\'s host read-back' if graphed else 'eager launches')
                             + ('' if world == 1 else '; per-layer gradient all-reduce buckets on a side stream, overlapped with the '
                                'backward' + (' (captured in the graph)' if graphed else '')),
                     'launch_mode': 'hipGraph' if graphed else 'eager',
                     'eager_ms_per_step': eager_ms, 'graph_capture_error': graph_error,
                     'grad_allreduce_buckets': len(tr.reducer.buckets) if tr.reducer else 0,
                     'grad_allreduce_overlap': overlap,
                     # n_gpus > 1: the same graphed step without the reducer, timed in this process (max over ranks), and
                     # single / data-parallel = the weak-scaling efficiency of the training step (x n_gpus = the speed-up)
                     'single_rank_ms_per_step': single_ms,
                     'scaling_efficiency': (single_ms / tms) if single_ms else None,
                     'speedup_over_one_gpu': (single_ms / tms * world) if single_ms else None,
                     'train_roofline': {'algorithmic_flops_per_step': 3.0 * fwd_flops,
                                        'achieved_TFLOPs': 3.0 * fwd_flops / (tms * 1e-3) / 1e12, 'mfma_peak_TFLOPs': 2500.0,
                                        'frac_of_mfma_peak': 3.0 * fwd_flops / (tms * 1e-3) / 1e12 / 2500.0,
                                        'algorithmic_bytes_per_step': 3 * step_bytes,
                                        'achieved_GBs': 3 * step_bytes / (tms * 1e-3) / 1e9,
                                        'frac_of_8TBs': 3 * step_bytes / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
            log(f'train step {train["ms_per_step"]:.2f} ms ({train["launch_mode"]})')
        out['train_step'] = train
        gc.collect()
        # ---- secondary figure: BASELINE configs[4], the sparse masked-denoise path (minecraft/sparse_diffusion.py): 64-frame clips of
        # 16x16 latents, codebook 8192, 512 context tokens per clip, dim 512 / 4 heads x 128 / depth 8 / mlp 1024, global batch 48 on
        # 8 GPUs = 6 clips per GPU; one full training step (position sampling, gather, corruption, forward, chunked 8192-way
        # linear + cross-entropy, backward, AdamW), eager launches.
        sparse = None
        if a.train_steps > 0 and not a.no_cone:
            from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
            from world_modelz_amd.train import SparseDenoiserTrainer
            del tr
            torch.manual_seed(43)
            sm = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).to(dev)
            st = SparseDenoiserTrainer(sm, 8192, num_context=512, lr=1e-4, warmup=500, distributed=world > 1)
            zs = torch.randint(0, 8192, (6, 64, 16, 16), generator=gen).to(dev)
            rs = torch.full((6,), 0.5)
            # (a data-parallel step whose capture failed above is not tried again here)
            sparse_graphed = not a.eager and (world == 1 or (backend == 'nccl' and train is not None and train['launch_mode'] == 'hipGraph'))
            if sparse_graphed and capture_agreed(st, zs):
                sparse_graphed = False
            for _ in range(3):
                st.train_step(zs, r=rs)
            barrier()
            s0 = time.perf_counter()
            nst = max(1, a.train_steps // 3)
            for _ in range(nst):
                st.train_step(zs, r=rs)
            torch.cuda.synchronize()
            barrier()
            sel = time.perf_counter() - s0
            if world > 1:
                t = torch.tensor([sel], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                sel = float(t.item())
            sparse = {'value': 6 * world * nst / sel, 'unit': 'clips/s', 'ms_per_step': sel / nst * 1e3, 'steps': nst,
                      'tokens_per_s': 6 * 512 * world * nst / sel,
                      'what': 'config 5 per GPU: 6 clips x 512 context tokens of 64x16x16 latents, codebook 8192, '
                              'VqSparseDiffusionModel dim 512 / 4x128 / depth 8 / mlp 1024, full training step ('
                              + ('one hipGraph replay per step)' if sparse_graphed else 'eager launches)'), 'launch_mode': 'hipGraph' if sparse_graphed else 'eager'}
            log(f'sparse (config 5) train step {sparse["ms_per_step"]:.2f} ms')
        out['sparse_step'] = sparse
    except Exception as e:  # noqa: BLE001
        import traceback
        out['secondary_error'] = ''.join(traceback.format_exception_only(type(e), e)).strip()[:500]
        log('secondary figure failed: ' + out['secondary_error'])
    if rank == 0 and not a.no_cpu_baseline:
        # the oracle timed on this host's cores, on rank 0 at EVERY world size (the other ranks wait in the barrier below; their
        # watchdogs stay armed)
        log(f'cpu baseline on {usable_cores()} threads')
        try:
            out['cpu_baseline'] = cpu_baseline(cfg, sd_cpu)
        except Exception as e:  # noqa: BLE001
            out['cpu_baseline'] = None
            out['cpu_baseline_error'] = f'{type(e).__name__}: {e}'[:300]
    elif rank == 0:
        out['cpu_baseline'] = None
    if world > 1:
        # Leave without tearing the communicator down by hand: trainers of this process hold captured hipGraphs with RCCL kernels
        # in them, and a destroy_process_group() that aborts would turn a measured run into a failed one.  Everything is drained
        # and every rank passes the barrier; the process exit frees the rest.
        torch.cuda.synchronize()
        barrier()
    if watchdog is not None:
        watchdog.cancel()                  # only now: a rank that hung above must take the others down with it, not strand them
    if not printed.acquire(blocking=False):
        time.sleep(3600)                   # the watchdog is printing / ending this rank
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
