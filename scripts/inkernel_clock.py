"""In-kernel clock of the three chip-filling kernels (MI355X_MICROARCH.md, DVFS item 6).

DIAGNOSTIC build only:  bash scripts/build_stamps.sh  (here)  then on the GPU box
    VPHO_HIP_LIB=scripts/_ab/libvpho_hip_stamps.so python scripts/inkernel_clock.py > gpurun_out/inkernel_clock.txt
Every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) once in front of and once behind its main loop; the stamps
go to a buffer of their own.  Each kernel is launched back to back on tensors of the predict step's own shapes (random data) for >= 2.5 s,
then the stamps of the last launch are read: clock = d memtime / d memrealtime x 100 MHz, median over workgroups.  Beside it the
launch's HIP-event time in that steady state and the TFLOP/s it amounts to, against the fp32-MFMA peak at 2.4 GHz (157.3) and at the
measured clock."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
sys.path.insert(0, ROOT)
assert 'stamps' in os.environ.get('VPHO_HIP_LIB', ''), 'run with VPHO_HIP_LIB=scripts/_ab/libvpho_hip_stamps.so (scripts/build_stamps.sh)'
import numpy as np
import torch
from vpho_amd import ops
from vpho_amd.model.pack import winograd_weights

SECONDS = 2.5
PEAK = 157.3


WORDS = 10


def stamps(name, slots):
    fn = getattr(ops.lib, f'vpho_diag_stamps_{name}')
    fn.restype = C.c_int
    buf = np.zeros(WORDS * slots, dtype=np.uint64)
    torch.cuda.synchronize()
    assert fn(buf.ctypes.data_as(C.c_void_p), C.c_int(slots), C.c_int(1)) == 0
    b = buf.reshape(slots, WORDS)
    return b[b[:, 7] > 0]


def timeline(b, clock_ghz):
    """the launch as its workgroups saw it: phases of a workgroup's life, and -- from the chip-wide 100 MHz counter -- how the workgroups of
    one CU follow each other and how much of the launch's wall time a CU has 0 / 1 / 2 workgroups inside their main loops"""
    t = b[:, :6].astype(np.float64)
    cyc_us = 1e-3 / clock_ghz                                       # microseconds per shader cycle
    ph = {'entry -> first fills requested': (t[:, 1] - t[:, 0]) * cyc_us, 'first fill wait': (t[:, 2] - t[:, 1]) * cyc_us,
          'main loop': (t[:, 3] - t[:, 2]) * cyc_us, 'epilogue': (t[:, 4] - t[:, 3]) * cyc_us}
    if (t[:, 5] > 0).all():                                         # an extra stamp inside the epilogue (score head: partial sums done, in front of the barrier)
        ph['epilogue up to its stamp'] = (t[:, 5] - t[:, 3]) * cyc_us
    r0, r1 = b[:, 6].astype(np.int64), b[:, 7].astype(np.int64)
    rb = b[:, 9].astype(np.int64)
    re = rb + (b[:, 8] >> np.uint64(32)).astype(np.int64)
    cu = ((b[:, 8] >> np.uint64(8)) & np.uint64(0xFFF)).astype(np.int64)        # xcc id (4 bits) | se / sh / cu id of HW_ID
    start = r0.min()
    wall = (r1.max() - start) / 100.0
    life = (r1 - r0) / 100.0
    out = [f'    workgroup life {np.median(life):.1f} us median (p05 {np.percentile(life, 5):.1f}, p95 {np.percentile(life, 95):.1f}); launch in-kernel wall (first entry -> last exit) {wall:.1f} us; '
           f'{len(np.unique(cu))} CUs seen']
    out.append('    phases of a workgroup (median / p95, us): ' + '; '.join(f'{k} {np.median(v):.2f} / {np.percentile(v, 95):.2f}' for k, v in ph.items()))
    gaps, cover = [], np.zeros(4)
    first_entry = []
    for c in np.unique(cu):
        m = cu == c
        o = np.argsort(r0[m])
        s0, s1 = r0[m][o], r1[m][o]
        first_entry.append((s0[0] - start) / 100.0)
        for i in range(len(s0)):
            ended = s1[:i][s1[:i] <= s0[i]]
            if len(ended) and i >= 2:
                gaps.append((s0[i] - ended.max()) / 100.0)
        ev = sorted([(x, 1) for x in rb[m]] + [(x, -1) for x in re[m]])
        depth, last = 0, start
        for x, d in ev:
            cover[min(depth, 3)] += x - last
            depth += d
            last = x
        cover[0] += r1.max() - last
    cover = cover / cover.sum()
    out.append(f'    first workgroup of a CU enters {np.median(first_entry):.2f} us after the launch\'s first (p95 {np.percentile(first_entry, 95):.2f}); '
               + (f'slot turnover (a workgroup exits -> the next one\'s first instruction on that CU) {np.median(gaps):.2f} us median, p95 {np.percentile(gaps, 95):.2f} ({len(gaps)} hand-overs)' if gaps else 'one round: no slot turnover'))
    out.append(f'    share of the launch\'s in-kernel wall a CU has 0 / 1 / 2 / >2 workgroups inside their main loops: {cover[0]:.3f} / {cover[1]:.3f} / {cover[2]:.3f} / {cover[3]:.3f}')
    return '\n'.join(out)


def run(label, name, fn, flop, slots, first_slot=0):
    stamps(name, 8)                                               # clear
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = 0.0
    while time.perf_counter() - t0 < SECONDS:
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 50
        n += 50
    b_all = stamps(name, slots)
    if first_slot:                                                # the launch's timeline origin stays its first workgroup's entry
        origin = int(b_all[:, 6].min())
        print(f'    (workgroups {first_slot} .. {slots - 1} of the launch; they enter {(int(b_all[first_slot:, 6].min()) - origin) / 100.0:.1f} us after the launch\'s first workgroup, '
              f'the last one exits at {(int(b_all[first_slot:, 7].max()) - origin) / 100.0:.1f} us; the others exit by {(int(b_all[:first_slot, 7].max()) - origin) / 100.0:.1f} us)')
    b = b_all[first_slot:]
    loop_real = (b[:, 8] >> np.uint64(32)).astype(np.float64)
    ok = loop_real > 0
    clk = (b[ok, 3] - b[ok, 2]).astype(np.float64) / loop_real[ok] * 0.1
    med = float(np.median(clk))
    tf = flop / ms / 1e9
    print(f'{label}\n    {n} back-to-back launches in {time.perf_counter() - t0:.1f} s; steady state {ms * 1e3:.1f} us/launch = {tf:.1f} TFLOP/s = {tf / PEAK:.3f} of the '
          f'2.4 GHz peak, {tf / (PEAK * med / 2.4):.3f} of the peak at the measured clock\n    in-kernel clock over the main loop: median {med:.3f} GHz  (p05 {np.percentile(clk, 5):.3f}, '
          f'p95 {np.percentile(clk, 95):.3f}; {len(clk)} workgroups stamped)\n' + timeline(b, med), flush=True)
    return med


def main():
    dev = 'cuda'
    g = torch.Generator().manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    print(f'# in-kernel clock, {torch.cuda.get_device_name(0)}; diagnostic build {os.environ["VPHO_HIP_LIB"]}; >= {SECONDS} s of back-to-back launches each')
    # --- bare MFMA loop (scripts/microbench/mfma_f32_peak_random.hip): what the fp32 matrix instruction alone holds
    exe = os.path.join(ROOT, 'scripts', '_ab', 'mfma_f32_clock')
    if os.path.exists(exe):
        print(subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout.strip(), flush=True)
    # --- conv_igemm_glds_kernel<128,128,4,2,false>: three layers of the predict step
    for (N, H, Cin, Cout, res, pers) in ((64, 32, 512, 128, False, '0'), (64, 16, 1024, 256, False, '0'), (64, 32, 128, 512, True, '0'), (64, 32, 128, 512, True, '1'), (64, 16, 256, 1024, True, '1')):
        os.environ['VPHO_CONV_PERS'] = pers
        x, w, b = rnd(N, H, H, Cin), rnd(Cout, Cin) * 0.05, rnd(Cout)
        r = rnd(N, H, H, Cout) if res else None
        tiles = (N * H * H // 128) * (Cout // 128)
        run(f'{"conv_igemm_pers_kernel<128,128,4,2> (stamps [1]..[4]: the FIRST tile of a workgroup; life: all its tiles)" if pers == "1" else "conv_igemm_glds_kernel<128,128,4,2,false>"}: 1x1 {Cin} -> {Cout} on {N} x {H} x {H}{" + residual" if res else ""} ({Cin // 32} k stages per tile, {tiles} tiles)',
            'conv', lambda: ops.conv2d_nhwc(x, w, b, out_slope=0.01, res=r), 2.0 * N * H * H * Cin * Cout, min(tiles, 65536))
    # --- conv_winograd_kernel
    for (N, H, Cc) in ((64, 64, 256), (64, 32, 128), (64, 16, 256)):
        x, w, b = rnd(N, H, H, Cc), rnd(Cc, 9 * Cc) * 0.02, rnd(Cc)
        u = winograd_weights(w)
        wgs = (N * H * H // 4 // 64) * (Cc // 64)
        os.environ['VPHO_WINO8'] = '0'
        run(f'conv_winograd_kernel: 3x3 {Cc} -> {Cc} on {N} x {H} x {H} ({Cc // 8} k stages per workgroup, {wgs} workgroups; TFLOP/s = EXECUTED Winograd products)',
            'wino', lambda: ops.conv3x3_winograd(x, u, b, out_slope=0.01), 2.0 * (N * H * H // 4) * 16 * Cc * Cc, min(wgs, 65536))
    # --- score_head_kernel (hand network, R = 6400 rows x 32 heads)
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict
    from vpho_amd.assets import synthetic_assets
    sd = synth_state_dict(vpho_net(synthetic_assets(0)), 1)
    net = ops.ScoreNet(sd, 'denoiser_hand', dev)
    feat, xx = rnd(64, 1024) * 0.3, rnd(6400, 96)
    key = [k for k in ops.PROF_CLASSES if 'head' in k][0]
    # exclusive time of the head kernel inside net.score (which also runs the pose encoder): HIP events of the C side
    for _ in range(3):
        net.score(feat, xx, 0.3, 100)
    torch.cuda.synchronize()
    ops.prof_enable(key, True)
    for _ in range(20):
        net.score(feat, xx, 0.3, 100)
    torch.cuda.synchronize()
    r = ops.prof_collect(key)
    ops.prof_enable(key, False)
    us = r['total_ms'] / r['launches'] * 1e3
    med = run('score_head_kernel (hand): 6400 rows x 32 heads, 128-row tiles (the launch also runs time embedding + pose encoder: see the exclusive figure below)',
              'head', lambda: net.score(feat, xx, 0.3, 100), r['flops'] / r['launches'], 1536)
    run('score_head_kernel (hand), the 32-row TAIL tiles of the same launch (256 workgroups behind the 1536 ordinary ones)', 'head',
        lambda: net.score(feat, xx, 0.3, 100), r['flops'] / r['launches'], 1792, first_slot=1536)
    tf = r['flops'] / r['total_ms'] / 1e9
    print(f'    score_head_kernel exclusive (HIP events around the kernel, 20 launches): {us:.1f} us = {tf:.1f} TFLOP/s = {tf / PEAK:.3f} of the 2.4 GHz peak, '
          f'{tf / (PEAK * med / 2.4):.3f} of the peak at the measured clock')


if __name__ == '__main__':
    main()
