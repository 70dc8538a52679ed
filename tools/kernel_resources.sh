#!/bin/bash
# Per-kernel register / LDS / spill figures of a built object (code object metadata):
#   tools/kernel_resources.sh neuron-descriptions_amd/csrc/build/gemm.o [name-filter]
# Runs on the build container (no GPU needed).
set -e
obj=$(realpath "$1"); filt=${2:-.}
tmp=$(mktemp -d); cp "$obj" $tmp/x.o; cd $tmp
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1
co=$(ls x.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if re.search(r'$filt',name):
        print('%-90s vgpr %s agpr %s sgpr %s spill_v %s scratch %s lds %s'%(name[:90],g('vgpr_count'),blk.split()[1] if blk.split() else '?',g('sgpr_count'),g('vgpr_spill_count'),g('private_segment_fixed_size'),g('group_segment_fixed_size')))
"
rm -rf $tmp
