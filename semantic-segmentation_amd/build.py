"""Build libdsaspp.so (gfx950) in-tree with hipcc.  Usable as a script or via build_library()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, 'include')
LIB = os.path.join(HERE, 'libdsaspp.so')
SOURCES = ['conv_mfma.hip', 'dgrad_group.hip', 'dgrad_wide.hip', 'conv_wide.hip', 'wgrad_wide.hip', 'wgrad3x3_strip.hip', 'conv3x3_halo.hip', 'dgrad3x3_fused.hip', 'conv1x1_stream.hip', 'wgrad_mfma.hip', 'wgrad_group.hip', 'elementwise.hip', 'loss.hip', 'optim.hip', 'mobilenet.hip', 'mobilenet_bwd.hip', 'stem_fused.hip', 'fp32_path.hip', 'eval_ops.hip', 'program.hip']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


# compile flags of the product build: part of the source hash, so that a change here rebuilds the library
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def source_hash(extra_flags=()):
    """sha256 over every file the library is compiled from (kernel sources, shared header, public header) and the
    compile flags, in a fixed order: what dsaspp_source_hash() of a library built from this tree returns.  A build with
    extra flags (diagnostic -D switches) has another hash than the product build of the same sources."""
    import hashlib
    h = hashlib.sha256()
    h.update(' '.join(FLAGS + list(extra_flags)).encode())
    for path in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'common.h'), os.path.join(INCLUDE, 'dsaspp.h')]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, 'rb').read())
    return h.hexdigest()[:16]


def library_hash():
    """The source hash baked into the built library, read without loading it (None if absent)."""
    if not os.path.exists(LIB):
        return None
    import re
    m = re.search(rb'DSASPP_SRC=([0-9a-f]{16})', open(LIB, 'rb').read())
    return m.group(1).decode() if m else None


def build_library(force=False, verbose=False, extra_flags=(), out=None):
    """Build the library.  With extra_flags (diagnostic builds, results possibly wrong on purpose) an explicit output
    path is required: the in-tree product library is only ever written by a build with the product flags."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    extra_flags = tuple(extra_flags)
    if extra_flags and out is None:
        raise ValueError('a build with extra flags %r needs an explicit output path (out=...); it must not replace %s'
                         % (extra_flags, LIB))
    lib = out or LIB
    sh = source_hash(extra_flags)
    # rebuilt whenever the sources or flags differ from what the library was built from (not by modification time: a
    # checkout or a copy changes times, and a stale prebuilt library would go unnoticed)
    if not force and out is None and library_hash() == sh:
        return LIB
    objs = []
    procs = []
    for src in srcs:
        obj = src[:-4] + ('.o' if out is None else '.%s.o' % sh)
        objs.append(obj)
        cmd = [_hipcc(), *FLAGS, '-I', INCLUDE, '-I', CSRC, *extra_flags, '-c', src, '-o', obj]
        if src.endswith('program.hip'):
            cmd.insert(-4, '-DDSASPP_SOURCE_HASH="%s"' % sh)
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        log, _ = p.communicate()
        if log.strip() and (verbose or p.returncode != 0):
            print(log)
        if p.returncode != 0:
            failed = True
            print('FAILED: %s' % src, file=sys.stderr)
    if failed:
        raise RuntimeError('hipcc failed building libdsaspp')
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, *objs]
    subprocess.check_call(cmd)
    if out is not None:
        for obj in objs:
            os.remove(obj)
    return lib


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
