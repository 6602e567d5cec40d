# oracle/_ref — the part of the REFERENCE that compiles from its own sources with plain g++.
# TEST INFRASTRUCTURE ONLY.  Sources are compiled where they lie under $(REF); nothing is copied.
#
#   _ref/lj_eos : /root/reference/test/MD/tools/lj_eos.cpp (Thol et al. LJTS rc=2.5 equation of state),
#                 built exactly as the reference's own test does (test/MD/tools/eos.sh:6: g++ -O3).
#                 It produces the E(rho), P(rho) the reference's end-to-end LJ test asserts
#                 (test/MD/test.bash:28-70).
#
# Everything else on the hot path is CUDA (nvcc, cuFFT, cuBLAS, CUB: SURVEY 8c) and is unbuildable here.
REF ?= /root/reference
CXX ?= g++

all: _ref/lj_eos

_ref/lj_eos: $(REF)/test/MD/tools/lj_eos.cpp
	@mkdir -p _ref
	$(CXX) -O3 -w $< -o $@

clean:
	rm -rf _ref
