"""Build libdsaspp.so (gfx950) in-tree with hipcc.  Usable as a script or via build_library()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, 'include')
LIB = os.path.join(HERE, 'libdsaspp.so')
SOURCES = ['conv_mfma.hip', 'dgrad_group.hip', 'conv3x3_halo.hip', 'conv1x1_stream.hip', 'wgrad_mfma.hip', 'wgrad_group.hip', 'elementwise.hip', 'loss.hip', 'optim.hip', 'mobilenet.hip', 'mobilenet_bwd.hip', 'fp32_path.hip', 'eval_ops.hip', 'program.hip']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def build_library(force=False, verbose=False, extra_flags=()):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, 'common.h'), os.path.join(INCLUDE, 'dsaspp.h'), os.path.abspath(__file__)]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    objs = []
    procs = []
    for src in srcs:
        obj = src[:-4] + '.o'
        objs.append(obj)
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', INCLUDE, '-I', CSRC,
               '-Wall', '-Wno-unused-function', *extra_flags, '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and (verbose or p.returncode != 0):
            print(out)
        if p.returncode != 0:
            failed = True
            print('FAILED: %s' % src, file=sys.stderr)
    if failed:
        raise RuntimeError('hipcc failed building libdsaspp')
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs]
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
