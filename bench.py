#!/usr/bin/env python
"""MILAN hot-path benchmark (driver contract: one JSON line on rank 0).

Workload (BASELINE.json `metric`): synthetic
4096 neurons x k=15 exemplars x 3x224x224 uint8 + {0,1} masks, ResNet-101
pyramid encoder -> attention-LSTM beam search (beam 50, length 15) -> LM (PMI)
rerank (lambda 0.2), seeded synthetic weights, V = 5004.  A "step" is one pass
of the hot path (milan_describe through the C ABI) over one chunk of `--chunk`
neurons; the default 16 steps x 256 neurons = the 4096-neuron workload.

What is timed:
  * `value`: K steps over uint8 exemplars ALREADY RESIDENT IN HBM (the bench
    contract), barrier + synchronize on both sides, max over ranks.
  * `pcie_inclusive` (same number of steps, in the same run): SURVEY 8(d)'s
    definition of the metric -- exemplars start as uint8 in PINNED HOST memory,
    travel H2D double-buffered on a side stream (milan_amd/ingest.py), and the
    top-1 token ids + scores are copied back to pinned host memory inside the
    timed region.
  * `f32_mode`: the same steps in the exact-fp32 MFMA mode (8 steps).
  * `other_configs`: SURVEY 8(d) config 1 (256 neurons, greedy, mi=False) and
    config 2 (1152 neurons, beam 16 + rerank) on the same GPU.
Rank r of N describes its own neurons (no data-path collective; weights are
broadcast from rank 0 over RCCL, results gathered at the end).  Default: every
rank runs `--steps` chunks ("scaling": "weak").  `--neurons-total T` fixes the
WHOLE-JOB size instead (rank r takes `partition(T, N, r, align=16)`; "scaling":
"strong"): `--gpus 8 --neurons-total 4096` is BASELINE's metric as quoted.

    python bench.py                      # 1 GPU
    python bench.py --gpus 8             # launches its own 8 ranks (torch.distributed.run)
    python bench.py --gpus 8 --neurons-total 4096   # the SCALE line: BASELINE's metric as quoted
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8   # the driver's form
"""
import argparse
import hashlib
import json
import os
import pathlib
import sys
import time

REPO = pathlib.Path(__file__).resolve().parent
for p in (REPO, REPO / 'neuron-descriptions_amd'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import torch  # noqa: E402

from milan_amd import hip, sharding, synthetic  # noqa: E402

# algorithmic GFLOP per neuron-description.
GFLOP_ENCODER = 233.97
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, f32-in MFMA
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 MFMA (not the 2:1-sparse figure)
PEAK_HBM_GBS = 8000.0  # HBM3E


def decoder_gflop(beam: int, k: int = 15, f: int = 3904, h: int = 512,
                  a: int = 512, e: int = 128, v: int = 5004,
                  t: int = 15) -> float:
    """SURVEY.md 8(d), hoisted form, per neuron."""
    rows = 1 + (t - 1) * beam
    return 2 * (k * f * a + 2 * f * h + rows *
                (h * a + k * a + h * f + k * f +
                 (e + f + h) * 4 * h + h * v)) / 1e9


def lm_gflop(beam: int, rerank: bool, e: int = 128, h: int = 512,
             v: int = 5004) -> float:
    if not rerank:
        return 0.0
    return 2 * beam * 16 * ((e + h) * 4 * h + 2 * h * 4 * h + h * v) / 1e9


def algorithmic_gflop(beam: int, rerank: bool, lm_rows=None) -> float:
    """SURVEY.md 8(d) per neuron-description.  `lm_rows` = (LSTM, vocabulary) fractions of the
    B x beam rows the rerank pass really multiplies (round 6: beams that share a prefix share
    their LM state -- the duplicate rows are redundant work like the per-row key projection the
    survey's hoisted form already excludes); None = every row, the survey's 9.55 GFLOP."""
    if lm_rows is None or not rerank:
        return GFLOP_ENCODER + decoder_gflop(beam) + lm_gflop(beam, rerank)
    e, h, v = 128, 512, 5004
    lm = 2 * beam * 16 * (((e + h) * 4 * h + 2 * h * 4 * h) * lm_rows[0] + h * v * lm_rows[1]) / 1e9
    return GFLOP_ENCODER + decoder_gflop(beam) + lm


def tail_row_fractions(masks: torch.Tensor):
    """Fractions of the last stage's 7 x 7 pixels the mask-aware tail (round 6) computes
    (accounting only): f0 = pixels with a non-zero level-4 mask weight (the central 2 x 2 of
    their 32 x 32 block holds a set mask pixel), f1 / f2 = their 3 x 3 neighbourhoods once /
    twice.  masks: (..., 224, 224) uint8."""
    m = (masks.reshape(-1, masks.shape[-2], masks.shape[-1]) != 0).float()
    c = (m[:, 15::32, 15::32] + m[:, 16::32, 15::32] + m[:, 15::32, 16::32] + m[:, 16::32, 16::32]) > 0
    f0 = c.float().unsqueeze(1)
    f1 = torch.nn.functional.max_pool2d(f0, 3, 1, 1)
    f2 = torch.nn.functional.max_pool2d(f1, 3, 1, 1)
    return float(f0.mean()), float(f1.mean()), float(f2.mean())


def lm_row_fractions(beam_tokens: torch.Tensor):
    """Fractions of the rerank pass's rows that are distinct work (accounting only, outside any
    timed region): at LM step t a row's LSTM state depends on its first t tokens only, its
    vocabulary product on the first t + 1 -- count the distinct prefixes per neuron."""
    n, b, t = beam_tokens.shape
    key = torch.zeros(n, b, dtype=torch.int64, device=beam_tokens.device)
    uniq = []
    for i in range(t):
        key = key * 1000003 + beam_tokens[:, :, i] + 1
        srt = key.sort(dim=1).values
        uniq.append(1 + (srt[:, 1:] != srt[:, :-1]).sum(dim=1))   # (n,)
    u = torch.stack(uniq, dim=1).double()                          # (n, t): distinct prefixes of length i + 1
    lstm = (1 + u[:, :t - 1].sum(dim=1)).mean() / (b * t)
    vocab = u.sum(dim=1).mean() / (b * t)
    return float(lstm), float(vocab)


def pooled_bytes_per_image(masks: torch.Tensor, width: int = 64) -> float:
    """Algorithmic bytes of the mask-weighted pooling (encoders.py:303-320):
    only feature pixels whose bilinear-downscaled mask weight is non-zero are
    read.  Accounting only (torch on the synthetic masks, outside any timed
    region): a level of stride s keeps pixel (y, x) iff the central 2x2 of its
    s x s block holds a set mask pixel."""
    m = masks.reshape(-1, masks.shape[-2], masks.shape[-1]) != 0
    total = 0.0
    for stride, ch in ((2, width), (4, 4 * width), (8, 8 * width),
                       (16, 16 * width), (32, 32 * width)):
        a, b = stride // 2 - 1, stride // 2
        nz = (m[:, a::stride, a::stride] | m[:, a::stride, b::stride] |
              m[:, b::stride, a::stride] | m[:, b::stride, b::stride])
        total += float(nz.sum()) * ch * 4
    return total / m.shape[0]


def cpu_baseline(sd, nv, beam, length, temperature, sample, gpu_describe=None,
                 gpu_encode_modes=None, gpu_config1=None):
    """Time the oracle (kind 'port') on the host cores on a bounded sample; with
    `gpu_describe`, also check the HIP path against it on that sample.

    `gpu_encode_modes(images, masks) -> {mode: features}`: the precision evidence --
    the first 8 neurons of the sample are also encoded by the oracle in float64, and
    every arithmetic (this run's mode, the exact-fp32 MFMA mode, the fp32 CPU oracle)
    is priced against that.  `gpu_config1`: SURVEY 8(d)'s own baseline definition
    (config 1: 256 neurons, greedy, mi=False, batch 16) on a bounded sample of it.
    """
    from oracle import milan_oracle as O
    # torch's CPU conv collapses when oversubscribed on many-core hosts (256
    # threads: 88 s/neuron measured in round 1); cap the pool and say so.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    images, masks = synthetic.exemplars(sample, k=15, size=224, seed=1)
    with torch.no_grad():
        # warm-up + calibration: ONE neuron, then size the sample to ~20 s
        t0 = time.perf_counter()
        O.encode(O.byte_to_float(images[:1]), masks[:1].float(), sd, chunk=15)
        t_one = time.perf_counter() - t0
        sample = max(1, min(sample, int(20.0 / max(t_one, 1e-3))))
        images, masks = images[:sample], masks[:sample]
        t0 = time.perf_counter()
        feats = O.encode(O.byte_to_float(images), masks.float(), sd, chunk=15)
        t_enc = time.perf_counter() - t0
        want = O.forward(feats, sd, nv, 'rerank', length, beam, temperature)
        t_all = time.perf_counter() - t0
    parity = None
    if gpu_describe is not None:
        # same inputs through the HIP path (untimed): the oracle as the checker
        got = gpu_describe(images, masks)
        tp = want['tokens'].shape[1]
        same = (got['tokens'][:, :tp].cpu() == want['tokens']).all(dim=1)
        # a differing description is a NEAR-TIE when the oracle's own top-2 gap of the
        # rerank score is below 1e-3 (scores are sums of <= 15 log-probs around -50)
        margin = want.get('rerank_margin')
        near = int(((~same) & (margin < 1e-3)).sum()) if margin is not None else None
        parity = {
            'neurons': sample,
            'identical_descriptions': int(same.sum()),
            'different_at_oracle_near_tie': near,
            'different_otherwise': (int((~same).sum()) - near
                                    if near is not None else None),
            'smallest_oracle_top2_margin': (float(margin.min())
                                            if margin is not None else None),
            'max_abs_feature_diff': float(
                (got['features'].cpu() - feats).abs().max()),
            'feature_scale': float(feats.abs().max()),
            'max_abs_score_diff_on_identical': float(
                (got['scores'].cpu() - want['scores'])[same].abs().max())
            if same.any() else None,
        }
    dtype_error = None
    if gpu_encode_modes is not None:
        n64 = min(8, sample)
        sd64 = {k: (v.double() if v.is_floating_point() else v)
                for k, v in sd.items()}
        with torch.no_grad():
            t1 = time.perf_counter()
            ref64 = O.encode(O.byte_to_float(images[:n64]).double(),
                             masks[:n64].double(), sd64, chunk=15)
            t64 = time.perf_counter() - t1
        scale = float(ref64.abs().max())

        def err(x):
            d = (x.double().cpu() - ref64).abs()
            return {'max_abs': float(d.max()), 'max_rel_to_scale': float(d.max()) / scale,
                    'rms_rel_to_scale': float(d.pow(2).mean().sqrt()) / scale}

        dtype_error = {
            'reference': (f'oracle encoder in float64 on the first {n64} neurons '
                          f'({n64 * 15} feature vectors of 3904, {t64:.1f} s)'),
            'feature_scale': scale,
            'oracle_fp32': err(feats[:n64]),
        }
        for mode, f in gpu_encode_modes(images[:n64], masks[:n64]).items():
            dtype_error[mode] = err(f)
    config1 = None
    if gpu_config1 is not None:
        # config 1 = 256 neurons, greedy, mi=False, reference batch 16,
        # all host cores (capped, see above), 1 warm-up batch + timed batches; the
        # full 256-neuron pass would take ~2 minutes, so 2 batches are timed
        im1, mk1 = synthetic.exemplars(48, k=15, size=224, seed=3)
        with torch.no_grad():
            def batch(lo):
                f = O.encode(O.byte_to_float(im1[lo:lo + 16]), mk1[lo:lo + 16].float(),
                             sd, chunk=15)
                return O.forward(f, sd, nv, 'greedy', length, 1, temperature, mi=False)
            batch(0)
            t2 = time.perf_counter()
            outs1 = [batch(16), batch(32)]
            t_c1 = time.perf_counter() - t2
        got1 = gpu_config1(im1[16:48], mk1[16:48])
        want1 = torch.cat([o['tokens'] for o in outs1])
        config1 = {
            'value': 32 / t_c1, 'unit': 'neuron-descriptions/sec', 'cores': cores,
            'kind': 'port',
            'sample': ('SURVEY 8(d) config 1 (alexnet conv5 width: 256 neurons, greedy, '
                       'mi=False, batch 16): 1 warm-up batch + 2 timed batches of 16 '
                       'of the 16 in a full pass'),
            'identical_greedy_descriptions_hip_vs_oracle': int(
                (got1['tokens'].cpu() == want1).all(dim=1).sum()),
            'neurons_checked': 32,
        }
    return {
        'parity_vs_oracle': parity,
        'dtype_error': dtype_error,
        'config1': config1,
        'value': sample / t_all,
        'unit': 'neuron-descriptions/sec',
        'cores': torch.get_num_threads(),
        'kind': 'port',
        'sample': (f'{sample} neurons x 15 exemplars x 224^2, same pipeline '
                   f'(beam {beam} + rerank), torch-CPU fp32 oracle on '
                   f'{torch.get_num_threads()} threads of {os.cpu_count()} '
                   f'cpus; encoder {t_enc:.1f}s of {t_all:.1f}s'),
    }


class ClockPowerSampler:
    """sclk and board power of one GPU sampled from sysfs (pp_dpm_sclk's current level,
    hwmon power1_average / power1_input) on a host thread during the timed region: the
    part is power-limited (profiles/r3_clocks_power.txt), so box-to-box spread of `value`
    is explainable from the record.  Reads two small files every 0.25 s; no rocm-smi
    process is started inside the timed region."""

    def __init__(self, device_index: int = 0):
        import glob
        self.samples = []
        self._stop = None
        self._thread = None
        cards = sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'))
        bus = None
        try:
            bus = torch.cuda.get_device_properties(device_index).pci_bus_id
        except Exception:  # noqa: BLE001 -- older torch: no pci_bus_id
            bus = None
        self.sclk_path = None
        for c in cards:
            dev = os.path.dirname(c)
            # sysfs name = domain:bus:device.function; compare the BUS field exactly
            # ('00:' also matches the domain prefix '0000:' of every card: ADVICE r4)
            parts = os.path.basename(os.path.realpath(dev)).split(':')
            if bus is None or (len(parts) >= 3 and parts[1].lower() == '%02x' % bus):
                self.sclk_path = c
                break
        if self.sclk_path is None and cards:
            self.sclk_path = cards[min(device_index, len(cards) - 1)]
        self.power_path = None
        if self.sclk_path:
            base = os.path.dirname(self.sclk_path)
            for name in ('power1_average', 'power1_input'):
                hits = glob.glob(os.path.join(base, 'hwmon', 'hwmon*', name))
                if hits:
                    self.power_path = hits[0]
                    break

    def _read(self):
        mhz = watts = None
        try:
            for line in open(self.sclk_path).read().splitlines():
                if line.rstrip().endswith('*'):
                    mhz = float(line.split(':')[1].strip().split('M')[0])
        except Exception:  # noqa: BLE001
            pass
        try:
            if self.power_path:
                watts = float(open(self.power_path).read()) / 1e6
        except Exception:  # noqa: BLE001
            pass
        return mhz, watts

    def start(self):
        import threading
        if not self.sclk_path:
            return
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append(self._read())
                self._stop.wait(0.25)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join()
        mhz = sorted(m for m, _ in self.samples if m)
        watts = sorted(w for _, w in self.samples if w)
        if not mhz and not watts:
            return None

        def stats(v):
            return None if not v else {'mean': sum(v) / len(v), 'median': v[len(v) // 2],
                                       'min': v[0], 'max': v[-1]}
        return {'samples': len(self.samples), 'sclk_mhz': stats(mhz),
                'board_power_w': stats(watts),
                'source': 'sysfs pp_dpm_sclk (current level) + hwmon power1_average, '
                          'every 0.25 s during the timed region, rank 0\'s GPU'}


# kernel-family key (milan_profile_read_kernels) -> rocprofv3 symbol
SYMBOLS = {'pp32_256': 'igemm_split16_pp32_kernel', 'pp32t_256': 'igemm_split16_pp32t_kernel<256>',
           'pp32_128': 'igemm_split16_pp32n_kernel<128> / pp32t_kernel<128>',
           'chain_wide': 'chain3_kernel', 'chain': 'chain_kernel', 'bneck': 'chain_kernel<.., CONV>',
           'stem': 'stem_fused_kernel', 'conv3': 'conv3_p64_kernel', 'f32': 'igemm_kernel',
           'f16': 'igemm_f16_pp32_kernel', 'split_other': 'igemm_split16_kernel / igemm_kernel<SPLIT>'}


def kernel_family(name):
    """rocprofv3 kernel name -> the kernel-family key of `roofline.by_kernel`
    (milan_profile_read_kernels): the ping-pong tile and its tap-inner form are ONE family."""
    if 'igemm_split16_pp32_kernel' in name or 'pp32t_kernel<256>' in name:
        return 'pp32_256'
    if 'pp32n_kernel<128>' in name or 'pp32t_kernel<128>' in name:
        return 'pp32_128'
    if 'igemm_f16_pp32' in name:
        return 'f16'
    if 'chain3_kernel' in name:
        return 'chain_wide'
    if 'chain_kernel' in name:
        # (template argument list ends in the CONV flag: the 3x3 conv in front, round 6)
        import re
        conv = re.search(r'chain_kernel<64, 8, \w+, \d+, \w+, \w+, \d+, true>', name)
        return 'bneck' if conv else 'chain'
    if 'stem_fused' in name:
        return 'stem'
    if 'conv3_p64' in name:
        return 'conv3'
    if 'igemm' in name:
        return 'split_other'
    return None


def measure_traffic_live(args):
    """HBM bytes per launch of the dominant GEMM kernel, measured NOW: two child runs of
    this script (one step, same chunk, same precision) under `rocprofv3 --kernel-trace
    --pmc FETCH_SIZE` and `... WRITE_SIZE` (PMC needs its own passes; FETCH_SIZE and
    WRITE_SIZE do not fit one pass on gfx950), read back from rocprofv3's database.
    Counter unit KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.
    Returns (bytes_per_launch, note) or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which('rocprofv3')
    if rocprof is None:
        return None, 'rocprofv3 is not on PATH'
    totals = {}
    with tempfile.TemporaryDirectory(dir='/tmp') as tmp:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, '--kernel-trace', '--pmc', counter, '-d', out, '-o', 'p',
                   '--', sys.executable, str(REPO / 'bench.py'), '--steps', '1',
                   '--warmup', '0', '--chunk', str(args.chunk), '--precision',
                   args.precision, '--also-f32-steps', '0', '--cpu-sample', '0',
                   '--no-profile', '--from-host-steps', '0', '--other-configs', '0',
                   '--fast-steps', '0', '--live-traffic', '0']
            env = dict(os.environ, TMPDIR='/tmp')
            try:
                r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True,
                                   text=True, timeout=600)
            except subprocess.TimeoutExpired:
                return None, f'rocprofv3 --pmc {counter} pass timed out'
            dbs = glob.glob(os.path.join(out, '**', '*.db'), recursive=True)
            if r.returncode != 0 or not dbs:
                lines = [ln for ln in r.stderr.splitlines()
                         if 'rocprofv3]' not in ln and 'simple_timer' not in ln and
                         'amdgpu.ids' not in ln]
                return None, (f'rocprofv3 --pmc {counter} pass failed '
                              f'(rc {r.returncode}): ' + ' | '.join(lines[-4:])[-400:])
            db = sqlite3.connect(dbs[0])
            q = ('select kernel_name, count(*), sum(value) from counters_collection '
                 'where counter_name = ? group by kernel_name')
            for name, launches, total in db.execute(q, (counter,)):
                fam = kernel_family(name)
                if fam is not None:
                    have = totals.setdefault(fam, {}).get(counter, (0, 0.0))
                    totals[fam][counter] = (have[0] + launches, have[1] + total)
    families = {}
    for fam, t in totals.items():
        if 'FETCH_SIZE' not in t or 'WRITE_SIZE' not in t:
            continue
        launches = t['FETCH_SIZE'][0]
        families[fam] = ((2 * t['FETCH_SIZE'][1] + t['WRITE_SIZE'][1]) * 1024 / launches,
                         launches)
    if not families:
        return None, 'no GEMM kernel found in the PMC databases'
    # (the caller picks the family its HIP-event timing found dominant)
    return families, (f'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / '
                      f'WRITE_SIZE (two child passes of one {args.chunk}-neuron step), '
                      'summed over the launches of the dominant kernel family; '
                      'FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, per launch')


def stage_report(stages, n_steps, n_images_per_step, neurons_per_step, beam,
                 rerank, split, pool_bytes_img, step_ms):
    """north_star: achieved fraction of the HBM / MFMA roofline per stage."""
    peak_tf = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    mfma_per_product = 3.0 if split else 1.0
    out = []

    def mfma(name, keys, label, flops_override=None):
        ms = sum(stages[k]['region_ms'] for k in keys)
        gemm_ms = sum(stages[k]['gemm_ms'] for k in keys)
        flops = sum(stages[k]['gemm_flops'] for k in keys)
        gbytes = sum(stages[k].get('gemm_bytes', 0.0) for k in keys)
        if flops_override is not None:
            flops = flops_override
        if ms <= 0:
            return
        ach = flops / (ms * 1e-3) / 1e12
        # the same stage against the other roof: algorithmic HBM bytes of its
        # GEMM launches (each operand once) over the stage time.  A stage is
        # labelled by the roof it sits closer to (MFMA issue = 3 instructions
        # per product in split mode).
        hbm_gbs = gbytes / (ms * 1e-3) / 1e9
        issue = ach * mfma_per_product / peak_tf
        entry = {
            'stage': name, 'what': label,
            'bound': 'hbm' if hbm_gbs / PEAK_HBM_GBS > issue else 'mfma',
            'ms_per_step': ms / n_steps,
            'ms_per_256_neurons': ms / n_steps * 256.0 / neurons_per_step,
            'gemm_ms_per_step': gemm_ms / n_steps,
            'achieved': ach, 'peak': peak_tf, 'unit': 'TFLOP/s',
            'frac': ach / peak_tf,
            'mfma_issue_frac': issue,
            'hbm_achieved_GBs': hbm_gbs,
            'hbm_frac': hbm_gbs / PEAK_HBM_GBS,
            'algorithmic_gemm_GB_per_step': gbytes / n_steps / 1e9,
            'share_of_step': ms / n_steps / step_ms,
        }
        out.append(entry)

    def hbm(name, key, label, bytes_per_image):
        ms = stages[key]['region_ms']
        if ms <= 0:
            return
        total = bytes_per_image * n_images_per_step * n_steps
        ach = total / (ms * 1e-3) / 1e9
        out.append({
            'stage': name, 'what': label, 'bound': 'hbm',
            'ms_per_step': ms / n_steps,
            'ms_per_256_neurons': ms / n_steps * 256.0 / neurons_per_step,
            'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
            'frac': ach / PEAK_HBM_GBS,
            'algorithmic_bytes_per_image': bytes_per_image,
            'share_of_step': ms / n_steps / step_ms,
        })

    hw = 224 * 224
    hbm('encoder.input', 'enc_input',
        'mask pyramid lists + u8 -> normalised NHWC input (u8 image + u8 mask '
        'read, 16 B/pixel written)', 3 * hw + hw + 16 * hw)
    # split mode: conv1 + bn1 + relu + maxpool are ONE launch (csrc/stem.hip) and the
    # stem_tail region is empty; the fp32 mode keeps the separate launches
    mfma('encoder.stem', ['enc_stem'],
         'conv1 7x7/2 + bn1 + relu + maxpool 3x3/2 as one launch (split mode; raw '
         'conv1 rows written only inside the mask bounding box) / conv1 alone as '
         'implicit GEMM (fp32 mode)')
    hbm('encoder.stem_tail', 'enc_stem_tail',
        'bn1 + relu + maxpool 3x3/2 (112^2x64 fp32 read, 56^2x64 written; fp32 mode '
        'only)', (112 * 112 * 64 + 56 * 56 * 64) * 4)
    for i in range(1, 5):
        mfma(f'encoder.layer{i}', [f'enc_layer{i}'],
             f'torchvision layer{i} convolutions (implicit GEMM; split mode: chained '
             f'expand -> reduce launches and register-resident 3x3 weights in layer1)'
             if i < 3 else f'torchvision layer{i} convolutions (implicit GEMM)')
    hbm('encoder.pool', 'enc_pool',
        'mask-weighted pooling of the five taps (only pixels under the mask '
        'are read)', pool_bytes_img + 3904 * 4)
    mfma('decoder.init', ['dec_init'],
         'hoisted key projection + init_state GEMMs')
    mfma('decoder.search', ['dec_search'],
         'T-step attention-LSTM search: GEMMs + attention / top-k / beam '
         'kernels; achieved = algorithmic decoder FLOPs / stage time '
         '(latency- and glue-bound, not at either roof)')
    if rerank:
        mfma('decoder.lm_rerank', ['dec_lm'],
             'LM scoring of every beam + PMI argmax; achieved = algorithmic '
             'LM FLOPs / stage time')
    return out


def rccl_version():
    try:
        return '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # torch built without it
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--chunk', type=int, default=640,
                    help='neurons per step (per GPU)')
    ap.add_argument('--neurons-total', type=int, default=0,
                    help='whole-job neuron count, sharded over the ranks in '
                    'batch-aligned blocks (strong scaling; overrides --steps: '
                    'every rank runs ceil(shard / chunk) steps).  0 = every '
                    'rank runs --steps chunks (weak scaling, the default)')
    ap.add_argument('--beam', type=int, default=50)
    ap.add_argument('--length', type=int, default=15)
    ap.add_argument('--temperature', type=float, default=0.2)
    ap.add_argument('--vocab', type=int, default=5000,
                    help='vocabulary tokens (V = vocab + 4 specials)')
    ap.add_argument('--strategy', default='rerank',
                    choices=['greedy', 'beam', 'rerank'])
    ap.add_argument('--cpu-sample', type=int, default=32,
                    help='neurons for the CPU baseline leg (0 = skip)')
    ap.add_argument('--no-profile', action='store_true',
                    help='do not bracket GEMM launches / stages with HIP events')
    ap.add_argument('--from-host-steps', type=int, default=-1,
                    help='steps of the PCIe-inclusive leg (pinned host uint8 -> '
                    'double-buffered H2D -> describe -> D2H of tokens + scores); '
                    '-1 = as many as the main run, 0 = skip')
    ap.add_argument('--fast-steps', type=int, default=8,
                    help='extra timed steps in the FAST mode (layer3 / layer4 on plain f16: '
                    'narrower than the reference, a reported extra, never `value`), with its '
                    'caption-flip rate against f32 on --fast-agreement neurons; 0 = skip')
    ap.add_argument('--fast-agreement', type=int, default=1024)
    ap.add_argument('--precision', default='split_f16',
                    choices=['split_f16', 'f32'],
                    help='split_f16: operands as (hi,lo) f16 pairs, 3 f16 '
                    'MFMAs per product, fp32 accumulate (fp32-GEMM-class error, '
                    'same parity suite); f32: exact fp32-in MFMA')
    ap.add_argument('--also-f32-steps', type=int, default=8,
                    help='extra timed steps in f32 mode, reported under '
                    '"f32_mode" (0 = skip)')
    ap.add_argument('--live-traffic', type=int, default=1,
                    help='1: measure roofline.traffic in this run (two extra child '
                    'passes under rocprofv3 --pmc, ~1 min each, single-GPU runs '
                    'only); 0: quote the committed offline summary')
    ap.add_argument('--other-configs', type=int, default=1,
                    help='1: also time SURVEY 8(d) configs 1 and 2 (single-GPU '
                    'runs only)')
    args = ap.parse_args()

    # `python bench.py --gpus N` without torchrun: this process becomes the
    # launcher of N ranks (torch.distributed.run on 127.0.0.1, a free port);
    # rank 0 of that job prints the JSON line.  Under torchrun (the driver's
    # launch line) RANK / WORLD_SIZE are set and this is one of the ranks.
    if args.gpus > 1 and not sharding.launched_by_torchrun():
        sys.exit(sharding.self_launch(str(pathlib.Path(__file__).resolve()),
                                      sys.argv[1:], args.gpus))

    rank, world, local = sharding.init_from_env(args.gpus)
    # one GPU per rank; `% device_count` only matters when several ranks share the
    # single GPU of a test box (collectives over gloo then, sharding.self_launch)
    n_dev = max(1, torch.cuda.device_count())
    device = torch.device('cuda', local % n_dev)
    torch.cuda.set_device(device)
    shared = -(-world // n_dev)  # ranks per GPU
    if shared > 1:
        # every rank holds its own activation workspace (0.24 GB per neuron of the
        # chunk) and exemplar pool: size the chunk to this rank's share of the HBM
        total = torch.cuda.get_device_properties(device).total_memory
        fit = int(0.6 * total / shared / 0.25e9) // 64 * 64
        args.chunk = min(args.chunk, max(64, fit))

    # roofline.traffic, measured live: two child passes of ONE step under rocprofv3
    # --pmc, run FIRST -- before this process holds its 150 GB of workspace
    live_traffic = None
    if world == 1 and args.live_traffic and not args.no_profile:
        live_traffic = measure_traffic_live(args)

    nv = args.vocab
    blocks = synthetic.RESNET_BLOCKS['resnet101']
    # Rank 0 builds the (synthetic) checkpoint; the others receive it by RCCL
    # broadcast over xGMI, as a real deployment would distribute milan-base.pth.
    sd = synthetic.milan_state_dict(nv + 4, 'resnet101', seed=0) \
        if rank == 0 else None
    sd_dev = sharding.broadcast_state_dict(sd, device, src=0)
    ctx = hip.Context(hip.make_dims(sd_dev, nv, blocks=blocks), sd_dev, device)
    ctx.set_precision(args.precision)
    del sd_dev

    strategy = {'greedy': hip.GREEDY, 'beam': hip.BEAM,
                'rerank': hip.RERANK}[args.strategy]
    beam = 1 if strategy == hip.GREEDY else args.beam
    rerank = strategy == hip.RERANK

    # this rank's steps: (global index of the first neuron, neurons)
    strong = args.neurons_total > 0
    if strong:
        lo, hi = sharding.partition(args.neurons_total, world, rank, align=16)
        # a shard smaller than the chunk runs as ONE launch of its own size (rounded
        # up to 64 so that every rank -- also a ragged last one -- warms up and times
        # the same launch geometry): `--gpus 8 --neurons-total 4096` = one 512-neuron
        # chunk per rank
        biggest = -(-args.neurons_total // world)
        args.chunk = min(args.chunk, max(64, -(-biggest // 64) * 64))
        starts = list(range(lo, hi, args.chunk))
        sizes = [min(args.chunk, hi - a) for a in starts]
    else:
        starts = [i * args.chunk for i in range(args.steps)]
        sizes = [args.chunk] * args.steps
    n_steps = len(sizes)
    my_neurons = sum(sizes)
    # distinct resident chunks (<= 6144 neurons = 23 GB of uint8); longer runs
    # cycle through them -- every step still does the full encode + decode work.
    # Weak mode: every rank has its own data (seed 1 + rank).  Strong mode:
    # ONE workload indexed by the global neuron number (the same on every rank,
    # so an N-rank run describes exactly the neurons of the 1-rank run).
    if strong:
        n_pool = min(-(-args.neurons_total // 16) * 16,
                     args.chunk * max(1, 6144 // args.chunk))
        seed = 1
    else:
        n_pool = args.chunk * min(max(1, n_steps), max(1, 6144 // args.chunk))
        seed = 1 + rank
    n_data = max(1, n_pool // args.chunk)
    # uint8 exemplars resident in HBM before the timed region starts
    images, masks = synthetic.exemplars(n_pool, k=15, size=224, seed=seed,
                                        device=str(device))

    def resident(start, size):
        a = start % n_pool
        if a + size <= n_pool:
            return images[a:a + size], masks[a:a + size]
        return (torch.cat([images[a:], images[:a + size - n_pool]]),
                torch.cat([masks[a:], masks[:a + size - n_pool]]))

    step_data = [resident(a, sz) for a, sz in zip(starts, sizes)]
    torch.cuda.synchronize()

    def chunk_of(i, size):
        if size == sizes[i % max(1, n_steps)]:
            return step_data[i % n_steps]
        return resident(starts[i % n_steps], size)

    def step(i, size=None, strat=None, bm=None):
        im, mk = chunk_of(i, args.chunk if size is None else size)
        # (check=False: no status read / stream synchronisation inside the timed
        # region; the status word accumulates and is read once after it)
        return ctx.describe(im, mk, strategy if strat is None else strat,
                            args.length, beam if bm is None else bm, False,
                            args.temperature, group_size=16, check=False)

    for i in range(args.warmup):
        step(i)
    sharding.barrier()
    torch.cuda.synchronize()
    if not args.no_profile:
        hip.profile_enable(True)
    sampler = ClockPowerSampler(device.index or 0) if rank == 0 else None
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    outs = [step(i, sizes[i]) for i in range(n_steps)]
    torch.cuda.synchronize()
    sharding.barrier()
    elapsed = time.perf_counter() - t0
    clocks_power = sampler.stop() if sampler else None
    # split-f16 fails loudly: bit 0 = a value hit the +-65504 clamp in ANY timed step
    status_flags = ctx.status(clear=True)
    gemm_ms = gemm_flops = gemm_launches = stages = kernels = None
    lm_rows = None
    tail_rows, tail_over_per_image = None, 0.0
    if not args.no_profile:
        gemm_ms, gemm_flops, gemm_launches = hip.profile_read()
        stages = hip.profile_read_stages()
        kernels = hip.profile_read_kernels()
        hip.profile_enable(False)
        # The library prices a launch by the rows it is SIZED for; since round 6 the number of
        # exemplars that hold work (non-empty mask) is known on the device only, so the encoder
        # launches' algorithmic FLOPs / bytes are scaled here by the fraction this workload has
        # (the synthetic set zeroes 1 mask in 97).
        n_live = n_img = 0
        for i in range(n_steps):
            has_work = step_data[i][1][:sizes[i]].flatten(2).amax(dim=2) > 0   # (neurons, k)
            n_live += int(has_work.sum())
            n_img += has_work.numel()
        live_frac = n_live / max(1, n_img)
        for fam in ('pp32_256', 'pp32t_256', 'pp32_128', 'split_other', 'chain', 'chain_wide',
                    'stem', 'conv3', 'bneck'):
            if fam in kernels and fam not in ('pp32_256',):   # (pp32_256 also holds the decoder / LM products)
                kernels[fam]['flops'] *= live_frac
                kernels[fam]['bytes'] *= live_frac
        # Round 6: the rerank pass multiplies only the rows that are distinct work (beams that
        # share a prefix share their LM state; the class counts stay on the device), while the
        # library prices its launches at the B x beam rows they are SIZED for.  The fractions
        # follow from the beams themselves: take the over-count out of the family that holds
        # the LM's products (N = 2048 and N = 5004: the 256-column ping-pong tile).
        if rerank and os.environ.get('MILAN_LM_DEDUP', '1') != '0' and args.precision != 'f32' \
                and outs and outs[0].get('beam_tokens') is not None:
            fr = [lm_row_fractions(o['beam_tokens']) for o in outs]
            w = [float(sz) for sz in sizes[:len(fr)]]
            lm_rows = (sum(f[0] * x for f, x in zip(fr, w)) / sum(w),
                       sum(f[1] * x for f, x in zip(fr, w)) / sum(w))
            e_, h_, v_ = 128, 512, nv + 4
            row_steps = float(my_neurons) * beam * args.length
            over = 2.0 * row_steps * (((e_ + h_) * 4 * h_ + 2 * h_ * 4 * h_) * (1 - lm_rows[0]) +
                                      h_ * v_ * (1 - lm_rows[1]))
            if 'pp32_256' in kernels and kernels['pp32_256']['flops'] > over:
                kernels['pp32_256']['flops'] -= over
        # ... and the mask-aware tail: the last two bottlenecks run at the pixels the level-4
        # pooling (and their 3x3 neighbourhoods) read; their launches are sized for all 49.
        tail_rows = None
        tail_over_per_image = 0.0
        if args.precision != 'f32' and (int(os.environ.get('MILAN_CHAIN', '127')) & 64):
            acc = [0.0, 0.0, 0.0]
            for i in range(n_steps):
                f = tail_row_fractions(step_data[i][1][:sizes[i]])
                for q in range(3):
                    acc[q] += f[q] * sizes[i]
            tail_rows = tuple(a / max(1, sum(sizes[:n_steps])) for a in acc)
            c1 = 49 * 2048 * 512
            c23 = 49 * (9 * 512 * 512 + 512 * 2048)
            tail_over_per_image = 2.0 * (c1 * (1 - tail_rows[1]) + c23 * (1 - tail_rows[0]) +
                                         c1 * (1 - tail_rows[2]) + c23 * (1 - tail_rows[1]))
            over = tail_over_per_image * 15 * my_neurons
            if 'pp32_256' in kernels and kernels['pp32_256']['flops'] > over:
                kernels['pp32_256']['flops'] -= over
            if stages is not None and stages['enc_layer4']['gemm_flops'] > over:
                stages['enc_layer4']['gemm_flops'] -= over
        # by rocprofv3 symbol (top_kernels below) ...
        kernels_by_symbol = {k: dict(v) for k, v in kernels.items()}
        # ... and by tile function: the ping-pong tile and its tap-inner form are ONE family
        if 'pp32t_256' in kernels:
            for key in ('ms', 'flops', 'launches', 'bytes'):
                kernels['pp32_256'][key] += kernels['pp32t_256'][key]
            del kernels['pp32t_256']
    rank_seconds = sharding.all_ranks(elapsed, device)
    rank_neurons = sharding.all_ranks(float(my_neurons), device)
    elapsed = sharding.max_over_ranks(elapsed, device)
    if os.environ.get('MILAN_BENCH_DEBUG'):  # per-rank checksums on stderr
        def h(t):
            return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12]
        for i, o in enumerate(outs):
            print(f'[rank {rank}] step {i} start {starts[i]} size {sizes[i]} '
                  f'images {h(step_data[i][0])} masks {h(step_data[i][1])} '
                  f'tokens {h(o["tokens"])} scores {h(o["scores"])}',
                  file=sys.stderr, flush=True)

    # ---- SURVEY 8(d) metric: pinned host uint8 in, tokens + scores on host --
    pcie = None
    host_steps = n_steps if args.from_host_steps < 0 else args.from_host_steps
    if host_steps > 0:
        from milan_amd import ingest
        # <= 4.8 GB of pinned host memory
        nh = min(host_steps, n_steps, max(1, 1280 // args.chunk))
        host = [tuple(t.cpu().pin_memory() for t in step_data[i])
                for i in range(nh)]
        hsizes = [sizes[i % nh] for i in range(host_steps)]
        tok_host = [torch.empty(sz, args.length, dtype=torch.long,
                                pin_memory=True) for sz in hsizes]
        sc_host = [torch.empty(sz, pin_memory=True) for sz in hsizes]

        def fetch(i):
            return host[i % nh]

        sharding.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i, (im, mk) in enumerate(
                ingest.ChunkPrefetcher(fetch, host_steps, device)):
            o = ctx.describe(im, mk, strategy, args.length, beam, False,
                             args.temperature, group_size=16, check=False)
            tok_host[i].copy_(o['tokens'], non_blocking=True)
            sc_host[i].copy_(o['scores'], non_blocking=True)
        torch.cuda.synchronize()
        sharding.barrier()
        e_host = sharding.max_over_ranks(time.perf_counter() - t2, device)
        pcie = (sum(hsizes), e_host)
        # same data, same kernels: the host leg must reproduce the resident run
        for i in range(min(host_steps, nh)):
            assert torch.equal(tok_host[i], outs[i]['tokens'].cpu()), i
        del host

    # secondary measurement in the exact-fp32 mode (same workload, fewer steps)
    f32_mode = None
    if args.precision != 'f32' and args.also_f32_steps > 0:
        ctx.set_precision('f32')
        step(0)
        sharding.barrier()
        torch.cuda.synchronize()
        hip.profile_enable(True)
        t1 = time.perf_counter()
        for i in range(args.also_f32_steps):
            step(i)
        torch.cuda.synchronize()
        sharding.barrier()
        e32 = sharding.max_over_ranks(time.perf_counter() - t1, device)
        ms32, _, n32 = hip.profile_read()
        hip.profile_enable(False)
        ctx.set_precision(args.precision)
        f32_mode = (e32, ms32, n32)

    # the FAST mode (plain-f16 layer3 / layer4), a reported extra: throughput of the same
    # steps + what it costs in captions against the exact-fp32 mode
    fast_mode = None
    if world == 1 and args.precision != 'f32' and args.fast_steps > 0:
        ctx.set_precision('f16')
        step(0)
        sharding.barrier()
        torch.cuda.synchronize()
        hip.profile_enable(True)
        t5 = time.perf_counter()
        for i in range(args.fast_steps):
            step(i)
        torch.cuda.synchronize()
        e16 = time.perf_counter() - t5
        ms16, _, n16 = hip.profile_read()
        k16 = hip.profile_read_kernels()
        st16 = hip.profile_read_stages()
        hip.profile_enable(False)
        fast_flags = ctx.status(clear=True)
        ctx.set_precision(args.precision)
        sys.path.insert(0, str(REPO / 'tools'))
        import precision_agreement
        agree = precision_agreement.agreement(ctx, args.fast_agreement, chunk=min(256, args.chunk),
                                              device=str(device)) \
            if args.fast_agreement > 0 else None
        fast_mode = (e16, ms16, n16, k16, st16, fast_flags, agree)

    # SURVEY 8(d) configs 1 and 2 on this GPU (same weights, same kernels)
    other = None
    if world == 1 and args.other_configs:
        other = []
        for name, total, strat, bm in (
                ('config1: 256 neurons (alexnet conv5 width), greedy, mi=False',
                 256, hip.GREEDY, 1),
                ('config2: 1152 neurons (alexnet all units), beam 16 + rerank',
                 1152, hip.RERANK, 16),
                # the metric's own size, exactly (the headline cycles 640-neuron chunks)
                (f'config4 exact: 4096 neurons ({4096 // args.chunk} x {args.chunk} + '
                 f'{4096 % args.chunk}), beam 50 + rerank', 4096, hip.RERANK, 50)):
            szs = [min(args.chunk, total - a)
                   for a in range(0, total, args.chunk)]
            step(0, szs[0], strat, bm)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for i, sz in enumerate(szs):
                step(i, sz, strat, bm)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t4
            g_alg = algorithmic_gflop(bm, strat == hip.RERANK)
            other.append({
                'config': name, 'neurons': total, 'steps': len(szs),
                'value': total / dt, 'unit': 'neuron-descriptions/sec',
                'ms': 1e3 * dt,
                'algorithmic_tflops': g_alg * total / dt / 1e3,
            })

    # final gather of the top-1 token ids + scores (section 8e); not timed
    tokens = torch.cat([o['tokens'] for o in outs]) if outs else \
        torch.empty(0, args.length, dtype=torch.long, device=device)
    scores = torch.cat([o['scores'] for o in outs]) if outs else \
        torch.empty(0, device=device)
    all_tokens, all_scores = sharding.gather_results(tokens, scores, dst=0)
    counts = sharding.max_over_ranks(float(n_steps), device)

    if rank != 0:
        sharding.finalize()
        return
    neurons = args.neurons_total if strong else n_steps * args.chunk * world
    assert all_tokens.shape[0] == neurons, (all_tokens.shape, neurons)
    value = neurons / elapsed
    # host-side caption reconstruction of the gathered top-1 tokens (section 8d:
    # timed and reported separately from `value`)
    from milan_amd import lang
    indexer = lang.Indexer(lang.Vocab(synthetic.vocab_tokens(nv)), None, True,
                           True, True, True, args.length)
    t3 = time.perf_counter()
    captions = indexer.reconstruct(all_tokens.cpu().tolist())
    reconstruct_ms = 1e3 * (time.perf_counter() - t3)
    assert len(captions) == all_tokens.shape[0]
    max_steps = int(counts)
    result = {
        'metric': 'neuron-descriptions/sec (whole node), 4096 neurons x k=15 '
                  'exemplars',
        'value': value,
        'unit': 'neuron-descriptions/sec',
        'n_gpus': world,
        'steps': max_steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / max(1, max_steps),
        'higher_is_better': True,
        'scaling': 'strong' if strong else 'weak',
        'vs_baseline': None,
        'dtype': ('f32' if args.precision == 'f32' else
                  'f32-equivalent: operands split into (hi,lo) f16 pairs (22 '
                  'significant bits), 3 f16 MFMAs per product, f32 accumulate'),
        'data': 'synthetic',
        'config': {
            'workload': (f'{neurons} neurons x k=15 x 3x224x224 u8 + masks '
                         f'({my_neurons} on rank 0); resnet101 pyramid encoder '
                         f'-> attention-LSTM {args.strategy} (beam {beam}, '
                         f'length {args.length}, lambda {args.temperature}); V='
                         f'{nv + 4}; synthetic seeded weights; inputs resident '
                         f'in HBM'),
            'neurons_per_step': args.chunk,
            'neurons_total': neurons,
            # (flat keys: the driver's parser truncates `workload` at 128 characters)
            'k': 15, 'image_size': 224, 'trunk': 'resnet101', 'strategy': args.strategy,
            'beam': beam, 'length': args.length, 'lambda': args.temperature, 'vocab': nv + 4,
            'precision': args.precision,
            'parallelism': f'neuron-sharded x{world}',
            'gathered_tokens': list(all_tokens.shape),
            'gathered_tokens_sha256': hashlib.sha256(
                all_tokens.cpu().numpy().tobytes()).hexdigest(),
        },
        'host_reconstruct_ms': reconstruct_ms,
        # status word after the timed steps (milan_status): 0 = no split-f16 value was
        # clamped, no non-finite input pixel
        'status_flags': status_flags,
        # which collective backend the ranks used (a SCALE record must be auditable)
        'dist_backend': (torch.distributed.get_backend()
                         if sharding.is_distributed() else 'none (single process)'),
        'rccl_version': rccl_version(),
        'ranks_per_gpu': shared,
        'clocks_power': clocks_power,
        # per rank (its own clock around its own steps): a straggler GPU is visible
        'per_rank': {
            'neurons': [int(x) for x in rank_neurons],
            'seconds': rank_seconds,
            'value': [n / t if t > 0 else 0.0
                      for n, t in zip(rank_neurons, rank_seconds)],
        },
    }
    result['per_rank']['value_min'] = min(result['per_rank']['value'])
    result['per_rank']['value_max'] = max(result['per_rank']['value'])
    # HBM bytes per launch of the dominant kernel: PMC counters need their own
    # rocprofv3 passes (tools/pmc_traffic.sh), so the figure is read from the
    # committed summary of those passes, not collected live.
    traffic_bytes, traffic_note = None, 'not collected (PMC needs separate rocprofv3 passes)'
    live_reason = None
    dom_family = max(kernels, key=lambda k: kernels[k]['ms']) if kernels else None
    if live_traffic is not None:
        families, traffic_note = live_traffic
        if families is None:
            live_reason, traffic_note = traffic_note, None
        elif dom_family in families:
            traffic_bytes = families[dom_family][0]
            traffic_note += f' ({dom_family}: {families[dom_family][1]} launches in the step)'
        else:
            live_reason, traffic_note = f'no PMC rows for the family {dom_family}', None
    for name in ([] if traffic_bytes is not None else
                 ['r5_hbm_traffic.json', 'r4_hbm_traffic.json', 'r3_hbm_traffic.json', 'r2_hbm_traffic.json', 'r1_hbm_traffic.json']):
        tpath = REPO / 'profiles' / name
        if args.precision != 'f32' and tpath.exists():
            with open(tpath) as f:
                tj = json.load(f)
            traffic_bytes = tj['traffic_gb_per_launch'] * 1e9
            traffic_note = (
                f"bytes per launch of {tj['dominant_kernel']} (FETCH_SIZE x2 "
                f"gfx950 correction + WRITE_SIZE), offline rocprofv3 --pmc "
                f"passes summarised in profiles/{tpath.name}" +
                (f"; not measured live: {live_reason}" if live_reason else ''))
            break
    if gemm_ms:
        # (the tail's skipped rows are not credited as work either)
        g_alg = algorithmic_gflop(beam, rerank, lm_rows) - tail_over_per_image * 15 / 1e9
        per_launch_flop = g_alg * 1e9 * my_neurons / gemm_launches
        avg_ms = gemm_ms / gemm_launches
        achieved = per_launch_flop / (avg_ms * 1e-3) / 1e12
        split = args.precision != 'f32'
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        # per stage: the decode stages are priced with SURVEY's algorithmic
        # FLOPs (what `achieved` above uses), the encoder stages with the
        # un-padded 2*M*N*K of their own launches (sums to 233.97 GFLOP/neuron)
        if stages is not None:
            stages['dec_search']['gemm_flops'] = (
                decoder_gflop(beam) * 1e9 * my_neurons -
                stages['dec_init']['gemm_flops'])
            stages['dec_lm']['gemm_flops'] = (
                algorithmic_gflop(beam, rerank, lm_rows) - GFLOP_ENCODER -
                decoder_gflop(beam)) * 1e9 * my_neurons
        pool_bytes = pooled_bytes_per_image(masks[:args.chunk])
        # the DOMINANT kernel priced on its own launches (milan_profile_read_kernels):
        # algorithmic FLOPs of its launches / their summed HIP-event time
        kname = {'pp32_256': 'igemm_split16_pp32_kernel + its tap-inner form igemm_split16_pp32t_kernel<256> (one tile function)', 'pp32_128':
                 'igemm_split16_pp32n_kernel<128> + igemm_split16_pp32t_kernel<128>', 'split_other':
                 'igemm_split16_kernel / igemm_kernel<SPLIT> (other split tiles)',
                 'f32': 'igemm_kernel (v_mfma_f32_32x32x2_f32)', 'chain': 'chain_kernel',
                 'chain_wide': 'chain3_kernel (layer3 expand -> reduce)',
                 'stem': 'stem_fused_kernel', 'conv3': 'conv3_p64_kernel',
                 'other': 'other'}
        dom = max(kernels, key=lambda k: kernels[k]['ms']) if kernels else None
        dk = kernels[dom] if dom else None
        dom_achieved = (dk['flops'] / (dk['ms'] * 1e-3) / 1e12) if dk and dk['ms'] > 0 \
            else achieved
        result['roofline'] = {
            'bound': 'mfma',
            'kernel': (kname.get(dom, dom) + (' (v_mfma_f32_32x32x16_f16, 3 per product)'
                                              if split else '')) if dom else None,
            # achieved / frac: the dominant kernel's own launches
            'achieved': dom_achieved,
            'peak': peak,
            'unit': 'TFLOP/s',
            'frac': dom_achieved / peak,
            'kernel_launches': int(dk['launches']) if dk else None,
            'kernel_avg_launch_ms': dk['ms'] / dk['launches'] if dk and dk['launches'] else None,
            'kernel_time_frac_of_step': dk['ms'] * 1e-3 / elapsed if dk else None,
            'kernel_algorithmic_GB_per_launch': dk['bytes'] / dk['launches'] / 1e9
            if dk and dk['launches'] else None,
            # every GEMM-class launch together (the round 1-4 definition of `frac`)
            'achieved_all_gemm': achieved,
            'frac_all_gemm': achieved / peak,
            # the matrix cores execute 3 f16 MFMA flops per algorithmic flop
            'mfma_issue_frac': (3 * dom_achieved / peak) if split else
            dom_achieved / peak,
            # the three launches that dominate the step BY ROCPROFV3 SYMBOL (the ping-pong tile's
            # two kernel names apart), for a flat parser: symbol, ms per step, TF-eq, algorithmic
            # GB per launch
            'top_kernels': [
                {'symbol': SYMBOLS.get(k, k), 'family': k, 'ms_per_step': v['ms'] / n_steps,
                 'launches_per_step': v['launches'] / n_steps,
                 'tflops_eq': v['flops'] / (v['ms'] * 1e-3) / 1e12,
                 'frac_of_peak': v['flops'] / (v['ms'] * 1e-3) / 1e12 / peak,
                 'algorithmic_GB_per_launch': v['bytes'] / max(1.0, v['launches']) / 1e9,
                 'algorithmic_TBs': v['bytes'] / (v['ms'] * 1e-3) / 1e12}
                for k, v in sorted(kernels_by_symbol.items(), key=lambda kv: -kv[1]['ms'])[:3]
                if v['ms'] > 0],
            'live_image_fraction': live_frac,
            # fractions of the B x beam rows the rerank pass multiplies (LSTM, vocabulary): beams
            # that share a prefix share their LM state.  `algorithmic_gflop_per_neuron` counts
            # the LM at these fractions; the survey's every-row figure is 263.49
            'lm_rows_multiplied_fraction': list(lm_rows) if lm_rows else None,
            # fractions of the last stage's pixels the last two bottlenecks are computed at: the
            # pixels the level-4 pooling reads, their 3x3 neighbourhoods once / twice
            'tail_rows_computed_fraction': list(tail_rows) if tail_rows else None,
            'survey_gflop_per_neuron': algorithmic_gflop(beam, rerank),
            'by_kernel': {k: {'ms_per_step': v['ms'] / n_steps,
                              'launches_per_step': v['launches'] / n_steps,
                              'tflops': v['flops'] / (v['ms'] * 1e-3) / 1e12,
                              'algorithmic_GBs': v['bytes'] / (v['ms'] * 1e-3) / 1e9}
                          for k, v in kernels.items() if v['ms'] > 0} if kernels else None,
            'traffic': traffic_bytes,
            'traffic_note': traffic_note,
            'launches': gemm_launches,
            'avg_launch_ms': avg_ms,
            'algorithmic_gflop_per_neuron': g_alg,
            'counted_gflop_per_neuron': gemm_flops / 1e9 / my_neurons,
            'gemm_time_frac_of_step': gemm_ms * 1e-3 / elapsed,
            'stages': stage_report(stages, n_steps, my_neurons * 15 / n_steps,
                                   args.chunk, beam, rerank, split, pool_bytes,
                                   1e3 * elapsed / n_steps)
            if stages is not None else None,
        }
    else:
        result['roofline'] = None
    if pcie is not None:
        n_host, e_host = pcie
        result['pcie_inclusive'] = {
            'value': n_host * world / e_host,
            'unit': 'neuron-descriptions/sec',
            'steps': host_steps,
            'frac_of_value': n_host * world / e_host / value,
            'note': 'SURVEY 8(d) metric: uint8 exemplars in pinned host memory '
                    '-> async H2D on a side stream, double buffered '
                    '(milan_amd/ingest.py) -> milan_describe -> top-1 tokens + '
                    'scores copied to pinned host memory; all inside the timed '
                    'region, same steps as `value`'}
    if f32_mode is not None:
        e32, ms32, n32 = f32_mode
        g_alg = algorithmic_gflop(beam, rerank)
        n32_neurons = args.also_f32_steps * args.chunk
        result['f32_mode'] = {
            'value': n32_neurons * world / e32,
            'unit': 'neuron-descriptions/sec',
            'steps': args.also_f32_steps,
            'roofline_achieved_tflops': g_alg * 1e9 * n32_neurons / (ms32 * 1e-3) / 1e12,
            'roofline_frac_of_f32_mfma_peak':
                g_alg * 1e9 * n32_neurons / (ms32 * 1e-3) / 1e12 /
                PEAK_F32_MFMA_TFLOPS,
        }
    if fast_mode is not None:
        e16, ms16, n16, k16, st16, fast_flags, agree = fast_mode
        g_alg = algorithmic_gflop(beam, rerank)
        n16_neurons = args.fast_steps * args.chunk
        fk = k16.get('f16', {})
        result['fast_mode'] = {
            'note': 'NOT the headline and not a default: layer3 / layer4 of the trunk on plain '
                    'f16 operands (11 significant bits, 1 MFMA per product, 2-byte '
                    'activations), everything else split_f16; narrower than the reference',
            'value': n16_neurons / e16,
            'unit': 'neuron-descriptions/sec',
            'steps': args.fast_steps,
            'speedup_vs_value': n16_neurons / e16 / value,
            'status_flags': fast_flags,
            'all_gemm_tflops': g_alg * 1e9 * n16_neurons / (ms16 * 1e-3) / 1e12,
            # its own kernel against the dense f16 roof (1 MFMA flop per algorithmic flop)
            'f16_kernel_tflops': fk['flops'] / (fk['ms'] * 1e-3) / 1e12 if fk.get('ms') else None,
            'f16_kernel_frac_of_f16_peak':
                fk['flops'] / (fk['ms'] * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS if fk.get('ms') else None,
            'f16_kernel_algorithmic_GBs': fk['bytes'] / (fk['ms'] * 1e-3) / 1e9 if fk.get('ms') else None,
            'stage_ms_per_256': {k: v['region_ms'] / n16_neurons * 256 for k, v in st16.items()
                                 if v['region_ms'] > 0},
            # caption-flip rate and score differences against the exact-fp32 mode (and the
            # same for the bench default, for scale)
            'agreement_vs_f32': agree,
        }
        result['fast_mode_value'] = result['fast_mode']['value']
        if agree:
            result['fast_mode_caption_flip_rate'] = agree['f16']['caption_flip_rate']
    # flat copies of the figures a record parser should not have to dig for
    if result.get('pcie_inclusive'):
        result['pcie_inclusive_value'] = result['pcie_inclusive']['value']
    if result.get('f32_mode'):
        result['f32_mode_value'] = result['f32_mode']['value']
        result['f32_mode_frac'] = result['f32_mode']['roofline_frac_of_f32_mfma_peak']
    if result.get('roofline') and result['roofline'].get('stages'):
        result['stage_ms_per_256'] = {
            st['stage']: st['ms_per_256_neurons'] for st in result['roofline']['stages']}
        result['roofline_frac_all_gemm'] = result['roofline']['frac_all_gemm']
    if other is not None:
        result['other_configs'] = other
    if world == 1 and args.cpu_sample > 0:
        sd = synthetic.milan_state_dict(nv + 4, 'resnet101', seed=0)

        def gpu_describe(images, masks):
            return ctx.describe(images, masks, strategy, args.length, beam,
                                False, args.temperature, want_features=True)

        def gpu_encode_modes(images, masks):
            out = {}
            for mode in dict.fromkeys((args.precision, 'f32', 'split_f16')):
                ctx.set_precision(mode)
                out[mode] = ctx.describe(images, masks, hip.GREEDY, 1, 1, False,
                                         args.temperature,
                                         want_features=True)['features']
            ctx.set_precision(args.precision)
            return out

        def gpu_config1(images, masks):
            return ctx.describe(images, masks, hip.GREEDY, args.length, 1, False,
                                args.temperature, group_size=16)

        result['cpu_baseline'] = cpu_baseline(
            sd, nv, beam, args.length, args.temperature, args.cpu_sample,
            gpu_describe if rerank else None, gpu_encode_modes, gpu_config1)
    else:
        result['cpu_baseline'] = None
    print(json.dumps(result), flush=True)
    sharding.finalize()


if __name__ == '__main__':
    main()
