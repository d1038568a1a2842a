#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
for r in 1 2; do for g in gbench_head gbench gbench_xnobar gbench_xnomagic gbench_xneither; do echo "== $g round $r"; step tools/bin/$g; done; done > gpurun_out/r03_gbench_ab9.txt 2>&1
python3 - <<'PY'
import re,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
cur=None
for l in open('gpurun_out/r03_gbench_ab9.txt'):
    m=re.match(r'== (\S+) round',l)
    if m: cur=m.group(1); continue
    m=re.match(r'(\S+)\s+N=.*?([\d.]+) us\s+([\d.]+) TFLOP',l)
    if m: acc[cur][m.group(1)].append(float(m.group(3)))
for g,d in acc.items(): print('%-16s'%g, '  '.join('%s %s'%(k,'/'.join('%.0f'%x for x in v)) for k,v in d.items()))
PY
