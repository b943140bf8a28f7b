#!/usr/bin/env python3
"""Build oracle/_ref/: the REFERENCE's own kernels, compiled from where they lie.

Needs /root/reference (only mounted in the build container); outputs go to oracle/_ref/ only
(git-ignored, but shipped to the GPU box with the snapshot like any other built artefact).
No reference source is copied: each .cu file is read, its CUDA launch syntax
`kernel<<<grid, block[, shm, stream]>>>(` is rewritten IN MEMORY to
`PVREF_LAUNCH(kernel, grid, block[, shm, stream])(` and the text is piped straight into g++
together with ref_shim/cuda_on_cpu.h (a CUDA-execution-model-on-CPU shim).  The reference's own
`cuda_utils.cuh` and `*.cuh` prototypes are #included in place through -I.

  libpvcnn_ref_cpu.so         built with -ffp-contract=fast -mfma: gcc contracts a*b+c exactly where
                              nvcc's default -fmad=true would (the oracle pins the same with fmaf())
  libpvcnn_ref_cpu_nofma.so   built with -ffp-contract=off: quantifies how much of the result depends
                              on contraction at all (tests/test_oracle_vs_ref.py reports it)
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = '/root/reference/modules/functional/src'
OUT = os.path.join(HERE, '_ref')
SHIM = os.path.join(HERE, 'ref_shim')
KERNEL_FILES = ['voxelization/vox.cu', 'interpolate/trilinear_devox.cu', 'interpolate/neighbor_interpolate.cu',
                'ball_query/ball_query.cu', 'grouping/grouping.cu', 'sampling/sampling.cu']
LAUNCH = re.compile(r'(\w+)\s*<<<(.*?)>>>\s*\(', re.S)


def compile_stdin(text, obj, flags, subdir):
    cmd = ['g++', '-x', 'c++', '-std=c++17', '-O2', '-fPIC', '-fvisibility=hidden', '-w', *flags,
           '-include', os.path.join(SHIM, 'cuda_on_cpu.h'),
           '-I', os.path.join(SHIM, 'include'), '-I', os.path.join(REF_SRC, subdir), '-I', REF_SRC,
           '-c', '-', '-o', obj]
    subprocess.run(cmd, input=text.encode(), check=True)


def build(variant, flags):
    objs = []
    for rel in KERNEL_FILES:
        text = open(os.path.join(REF_SRC, rel)).read()
        text, n = LAUNCH.subn(r'PVREF_LAUNCH(\1, \2)(', text)
        assert n > 0, f'no kernel launch found in {rel}'
        obj = os.path.join(OUT, f'{variant}_{os.path.basename(rel)}.o')
        compile_stdin(text, obj, flags, os.path.dirname(rel))
        objs.append(obj)
    for src in ['runtime.cpp', 'ref_api.cpp']:
        obj = os.path.join(OUT, f'{variant}_{src}.o')
        compile_stdin(open(os.path.join(SHIM, src)).read(), obj, flags + ['-I', SHIM], '.')
        objs.append(obj)
    lib = os.path.join(OUT, f'libpvcnn_ref_cpu{"" if variant == "fma" else "_" + variant}.so')
    subprocess.run(['g++', '-shared', '-o', lib, *objs], check=True)
    for o in objs:
        os.remove(o)
    return lib


def main():
    if not os.path.isdir(REF_SRC):
        print(f'{REF_SRC} not present: skipping oracle/_ref (prebuilt files, if any, are kept)')
        return 0
    os.makedirs(OUT, exist_ok=True)
    print(build('fma', ['-mfma', '-ffp-contract=fast']))
    print(build('nofma', ['-ffp-contract=off']))
    return 0


if __name__ == '__main__':
    sys.exit(main())
