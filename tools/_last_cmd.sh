cd /tmp; rm -rf q2drun; mkdir q2drun; cd q2drun
for scheme in quasi2D true2D; do
cat > data.main <<EOF
boxSize 64 64
numberSteps 2000
printSteps 500
dt 0.01
relaxSteps 0
viscosity 1
temperature 1
hydrodynamicRadius 1
scheme $scheme
numberParticles 4096
loadParticles 0
output pos.$scheme
EOF
timeout 60 $GRAFT_REPO_ROOT/tools/_build/refrun/q2D data.main > out.$scheme 2> err.$scheme; echo "$scheme rc=$? $(wc -l < pos.$scheme 2>/dev/null) lines; $(grep -i 'error\|exception\|what' err.$scheme | tail -2 | cut -c1-200)"
python3 - <<P
import numpy as np
rows=[l.split() for l in open("pos.$scheme") if not l.startswith("#")]
a=np.array([[float(x) for x in r[:3]] for r in rows if len(r)>=3])
print("$scheme", a.shape, "finite", np.isfinite(a).all(), "x range", a[:,0].min(), a[:,0].max())
P
done
head -3 pos.quasi2D
