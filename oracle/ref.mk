# oracle/_ref — the part of the REFERENCE that compiles from its own sources with plain g++.
# TEST INFRASTRUCTURE ONLY.  Sources are compiled where they lie under $(REF); nothing is copied.
#
#   _ref/lj_eos : /root/reference/test/MD/tools/lj_eos.cpp (Thol et al. LJTS rc=2.5 equation of state),
#                 built exactly as the reference's own test does (test/MD/tools/eos.sh:6: g++ -O3).
#                 It produces the E(rho), P(rho) the reference's end-to-end LJ test asserts
#                 (test/MD/test.bash:28-70).
#   _ref/bravais_dump : oracle/ref_drivers/bravais_dump.c (ours: main() only) around the reference's lattice generator
#                 src/third_party/bravais/bravais.h (plain C, what utils/InitialConditions.cuh:17-32 `initLattice` calls); it
#                 produces tests/golden/bravais_lattices.npz, the pin of include/uammd/utils/InitialConditions.cuh and
#                 uammd_amd/initial_conditions.py.
#
# Everything else on the hot path is CUDA (nvcc, cuFFT, cuBLAS, CUB: SURVEY 8c) and is unbuildable here.
REF ?= /root/reference
CXX ?= g++

all: _ref/lj_eos _ref/bravais_dump

_ref/lj_eos: $(REF)/test/MD/tools/lj_eos.cpp
	@mkdir -p _ref
	$(CXX) -O3 -w $< -o $@

_ref/bravais_dump: ref_drivers/bravais_dump.c $(REF)/src/third_party/bravais/bravais.h
	@mkdir -p _ref
	gcc -O1 -w -I$(REF)/src/third_party/bravais $< -o $@ -lm

clean:
	rm -rf _ref
