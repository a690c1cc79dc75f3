#!/usr/bin/env python
"""Static check of the hand-scheduled kernels' ISA (no GPU needed): compiles each source to gfx950 assembly and reports,
per kernel, the register count, the spill count, and -- for every loop that contains MFMAs -- the scratch accesses and the
compiler-inserted `s_waitcnt vmcnt(0)` inside it.  In a kernel that feeds its operands by LDS-DMA and waits for them by a
hand-counted vmcnt, either of the two inside the K-loop drains the DMA queue once per iteration (a scratch reload is
waited for with vmcnt(0), because the compiler's own count does not include the DMAs).

    python tools/check_isa.py [source.hip ...]        (default: the DMA-fed kernels)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'semantic-segmentation_amd', 'csrc')
DEFAULT = ['dgrad_wide.hip', 'conv_wide.hip', 'wgrad_wide.hip', 'wgrad3x3_strip.hip', 'dgrad_group.hip', 'wgrad_group.hip', 'conv1x1_stream.hip', 'dgrad3x3_fused.hip', 'conv3x3_halo.hip']


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.strip().split('\n')))
    except Exception:
        return {n: n for n in names}


def check(src):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'k.s')
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', os.path.join(ROOT, 'include'), '-I', CSRC,
               '-Wno-unused-command-line-argument', '--cuda-device-only', '-S', src, '-o', asm]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(asm).read()
    meta = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', text):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    rows = []
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm', text, re.S | re.M):
        name, body = m.group(1), m.group(2).split('\n')
        labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'\.LBB\d+_\d+:', l)}
        loops = []
        for i, l in enumerate(body):                         # a backward branch closes a loop
            b = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if b and b.group(1) in labels and labels[b.group(1)] <= i:
                loops.append((labels[b.group(1)], i))
        worst = None
        for a, b in loops:
            seg = body[a:b + 1]
            nm = sum('v_mfma' in x for x in seg)
            if nm == 0:
                continue
            inner = not any(a < a2 and b2 < b and sum('v_mfma' in x for x in body[a2:b2 + 1]) for a2, b2 in loops)
            if not inner:
                continue
            sc = sum('scratch_' in x for x in seg)
            v0 = sum(bool(re.search(r's_waitcnt vmcnt\(0\)', x)) and 'lgkmcnt' not in x for x in seg)
            cand = (nm, sc, v0)
            if worst is None or (sc + v0, nm) > (worst[1] + worst[2], worst[0]):
                worst = cand
        rows.append((name, meta.get(name, (None, None)), len([1 for a, b in loops]), worst))
    return rows


def main():
    srcs = sys.argv[1:] or [os.path.join(CSRC, s) for s in DEFAULT]
    print('%-78s %5s %6s   %s' % ('kernel', 'vgprs', 'spills', 'innermost MFMA loops: worst (MFMAs, scratch ops, vmcnt(0) waits)'))
    for src in srcs:
        rows = check(src)
        names = demangle([r[0] for r in rows])
        for name, (vg, sp), nloops, worst in rows:
            short = re.sub(r'\(anonymous namespace\)::', '', names[name])
            short = re.sub(r'\(.*$', '', short)
            print('%-78s %5s %6s   %s' % (short[:78], vg, sp, worst if worst else '-'))


if __name__ == '__main__':
    main()
