"""Diagnostic: register, scratch and LDS figures of the kernels in a built library (from the code
objects' metadata notes).  Usage: python tools/kernel_resources.py [lib.so] [name filter]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_spills import code_objects, READELF
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bayesian_optimization_amd', 'liboei_hip.so')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for obj in code_objects(so):
    with tempfile.NamedTemporaryFile(suffix='.co') as tf:
        tf.write(obj); tf.flush()
        out = subprocess.run([READELF, '--notes', tf.name], stdout=subprocess.PIPE).stdout.decode()
    for blk in out.split('- .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk)
        if not name or flt not in name.group(1):
            continue
        dem = subprocess.run(['c++filt', name.group(1)], stdout=subprocess.PIPE).stdout.decode().strip()
        g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, '?'])[1]
        print('%-60s agpr %s vgpr %s sgpr %s scratch %s B  sgpr_spill %s vgpr_spill %s' % (
            dem[:60], blk.split()[0], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'),
            g('sgpr_spill_count'), g('vgpr_spill_count')))
