"""Build check: looks for a code-generation pattern in the built library that produced wrong results.

When a kernel needs more scalar registers than exist, the compiler parks them in lanes of vector
registers ("SGPR spill to VGPR lane", v_writelane/v_readlane).  If the kernel is also out of
vector registers, those carrier registers are themselves written to scratch memory inside
whole-wave sections (s_or_saveexec_b64 ..., -1 / scratch_store / restore exec).  The inlined
Q = 32 solver instance built that way computed wrong iterates, while the same source built with
-mllvm -amdgpu-spill-sgpr-to-vgpr=false was correct.  The library therefore must not contain such
sections beyond the save/restore pair at a function's entry and exit.

    python tools/check_spills.py [path/to/liboei_hip.so]     exit status 1 if the pattern is found
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(so_path):
    """gfx950 code objects embedded in the .hip_fatbin section (one uncompressed bundle per
    translation unit)."""
    out = subprocess.run([READELF, '-S', '--wide', so_path], stdout=subprocess.PIPE, check=True).stdout.decode()
    m = re.search(r'\.hip_fatbin\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)', out)
    if not m:
        raise RuntimeError('no .hip_fatbin section in ' + so_path)
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    with open(so_path, 'rb') as fh:
        fh.seek(off)
        blob = fh.read(size)
    objs = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from('<Q', blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tsize].decode()
            p += 24 + tsize
            if 'gfx950' in triple and esize:
                objs.append(blob[pos + eoff:pos + eoff + esize])
        pos = blob.find(MAGIC, pos + 1)
    if not objs:
        raise RuntimeError('no gfx950 code object found (compressed bundle?)')
    return objs


def scan(so_path):
    """{function: count of whole-wave scratch sections} for functions with more than the
    entry/exit pair."""
    bad = {}
    total = 0
    for oi, obj in enumerate(code_objects(so_path)):
        with tempfile.NamedTemporaryFile(suffix='.co') as tf:
            tf.write(obj)
            tf.flush()
            dis = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', tf.name], stdout=subprocess.PIPE,
                                 check=True).stdout.decode().split('\n')
        cur, prev_wwm = None, False
        for line in dis:
            m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
            if m:
                cur, prev_wwm = 'code object %d: %s' % (oi, m.group(1)), False
                total += 1
                continue
            ins = line.split('//')[0].strip()
            if prev_wwm and ('scratch_store' in ins or 'scratch_load' in ins):
                bad[cur] = bad.get(cur, 0) + 1
            prev_wwm = ins.startswith('s_or_saveexec_b64') and ins.rstrip().endswith('-1')
    return {k: v for k, v in bad.items() if v > 2}, total


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bayesian_optimization_amd', 'liboei_hip.so')
    bad, total = scan(so)
    print('%d functions scanned in %s' % (total, so))
    for k, v in sorted(bad.items()):
        print('carrier registers of scalar spills written to scratch %d times in %s' % (v, k))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
