#!/bin/bash
# dev: register / LDS use of the kernels in an object file or shared library (code-object metadata); usage: tools/kres.sh file [name-filter]
f=$1; pat=${2:-.}
d=$(mktemp -d); cp $f $d/in; cd $d
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading in > /dev/null 2>&1
co=$(ls | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $co | python3 -c "
import re,sys
txt=sys.stdin.read()
for k in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    name=re.search(r'\.name:\s+(\S+)',k).group(1)
    g=lambda f:(re.search(r'\.%s:\s+(\d+)'%f,k) or [0,'?'])[1]
    if re.search(r'$pat',name): print('%-90s vgpr %4s agpr %3s sgpr %3s lds %6s scratch %s'%(name[:90],g('vgpr_count'),re.match(r':\s+(\d+)',k).group(1),g('sgpr_count'),g('group_segment_fixed_size'),g('private_segment_fixed_size')))
"
rm -rf $d
