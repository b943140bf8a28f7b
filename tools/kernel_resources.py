#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of the built library, from the code objects' own metadata.

libpvcnn_hip.so embeds one clang offload bundle per translation unit; each bundle holds the gfx950 code object (an ELF whose
NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count, .sgpr_count, .vgpr_spill_count, .sgpr_spill_count,
.private_segment_fixed_size (scratch bytes per lane) and .group_segment_fixed_size (static LDS)).  This tool unbundles them by hand
(the bundle format is a 24-byte magic, an entry table of (offset, size, triple)) and reads the notes with llvm-readelf.

usage: kernel_resources.py [path/to/libpvcnn_hip.so] [--json]        (also imported by tests/test_kernel_resources.py)
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, 'pvcnn_amd', 'csrc', 'libpvcnn_hip.so')
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(path):
    """-> list of bytes: every amdgcn code object embedded in `path`."""
    data = open(path, 'rb').read()
    out = []
    for m in re.finditer(MAGIC, data):
        base = m.start()
        n, = struct.unpack_from('<Q', data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from('<QQQ', data, pos)
            triple = data[pos + 24:pos + 24 + idlen].decode()
            pos += 24 + idlen
            if 'amdgcn' in triple and size:
                out.append(data[base + off:base + off + size])
    return out


def kernels(path=DEFAULT_LIB):
    """-> {demangled-ish kernel symbol: {vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, scratch_bytes, lds_bytes}}"""
    res = {}
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(blob)
            f.flush()
            text = subprocess.run([READELF, '--notes', f.name], capture_output=True, text=True, check=True).stdout
        # the YAML is regular: one list item ('- .field: ...') per kernel
        for chunk in re.split(r'\n\s*- \.', '\n' + text):
            sym = re.search(r'\.symbol:\s*\'?([^\s\']+)', chunk)
            if not sym or '.vgpr_count' not in chunk:
                continue
            g = lambda k: int(re.search(r'\.' + k + r':\s*(\d+)', chunk).group(1)) if re.search(r'\.' + k + r':\s*(\d+)', chunk) else 0
            res[sym.group(1).replace('.kd', '')] = {
                'vgpr_count': g('vgpr_count'), 'sgpr_count': g('sgpr_count'), 'vgpr_spill_count': g('vgpr_spill_count'),
                'sgpr_spill_count': g('sgpr_spill_count'), 'scratch_bytes': g('private_segment_fixed_size'),
                'lds_bytes': g('group_segment_fixed_size'), 'max_threads': g('max_flat_workgroup_size')}
    return res


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.splitlines()))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    table = kernels(args[0] if args else DEFAULT_LIB)
    if '--json' in sys.argv:
        print(json.dumps(table, indent=1, sort_keys=True))
    else:
        names = demangle(list(table))
        print(f'{len(table)} kernels')
        for k, v in sorted(table.items(), key=lambda kv: (-kv[1]['vgpr_spill_count'], -kv[1]['scratch_bytes'], kv[0])):
            print(f"vgpr {v['vgpr_count']:4d} sgpr {v['sgpr_count']:4d} spill v{v['vgpr_spill_count']:3d} s{v['sgpr_spill_count']:3d} "
                  f"scratch {v['scratch_bytes']:5d} B  lds {v['lds_bytes']:6d} B  {names[k][:150]}")
