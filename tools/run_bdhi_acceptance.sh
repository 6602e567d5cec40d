#!/bin/bash
# usage (GPU box): tools/run_bdhi_acceptance.sh [Lanczos|Cholesky] [nsteps]  — the reference's test/BDHI/Lanczos_Cholesky pipeline (its program,
# built against include/uammd, piped into its own checker process.cpp): columns r |f - f_theo| / f_theo |g - g_theo| / g_theo
set -e
M=${1:-Lanczos}; NS=${2:-10}
D=$(mktemp -d); cd $D
cat > data.main <<EOD
N              5000
boxSize	       4 4 4
radius_min     0.38173
radius_max     1.89538
outfile	       /dev/stdout
temperature    0
viscosity      1.2131
dt	       10
tolerance      1e-8
nsteps	       $NS
printSteps     1
mode	       $M
EOD
$GRAFT_REPO_ROOT/examples/_build/ref_test_BDHI 2> run.log | $GRAFT_REPO_ROOT/examples/_build/ref_process_bdhi > dev.txt
python3 - <<PY
import numpy as np
d = np.loadtxt("dev.txt")
print("$M", d.shape[0], "pairs; f deviation max %.2e p99 %.2e; g deviation max %.2e p99 %.2e median %.2e" % (d[:,1].max(), np.quantile(d[:,1],0.99), d[:,2].max(), np.quantile(d[:,2],0.99), np.median(d[:,2])))
w = d[:,2] > 1e-2
print("g > 1e-2:", w.sum(), "pairs, r range", d[w,0].min() if w.any() else None, d[w,0].max() if w.any() else None)
PY
