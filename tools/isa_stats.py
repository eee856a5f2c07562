"""Per-kernel register / spill / occupancy / instruction-mix summary of a gfx950 assembly file (hipcc -save-temps=obj)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
names = re.findall(r'^(_Z\S+):\s*; @', txt, re.M)
blocks = re.split(r'^_Z\S+:\s*; @.*$', txt, flags=re.M)
for n, b in zip(names, blocks[1:]):
    dem = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    if pat and pat not in dem:
        continue
    g = lambda k: (re.search(r'; %s: (\d+)' % k, b) or [None, '?'])[1]
    print(dem[:110])
    print('    NumVgprs', g('NumVgprs'), 'NumAgprs', g('NumAgprs'), 'Scratch', g('ScratchSize'), 'Occupancy', g('Occupancy'),
          '| mfma', len(re.findall(r'\bv_mfma', b)), 'exp', len(re.findall(r'\bv_exp_f32', b)), 'ds_read', len(re.findall(r'\bds_read', b)),
          'scratch_ld/st', len(re.findall(r'\bscratch_(load|store)', b)))
