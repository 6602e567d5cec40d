"""The C++14 mirror of the reference's host interface (include/uammd/): compile + link here, run on the GPU."""
import os
import subprocess

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")
PROGS = ["custom_potential", "ibm_library_mode", "bd_readme", "lj_benchmark", "fcm_selfmobility", "pse_selfmobility", "poisson_two_charges", "checkpoint", "quasi2d_selfmobility", "particle_group", "custom_transverser", "module_lifetime"]


def _make():
    r = subprocess.run(["make", "-C", EX], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_examples_compile_and_link_with_plain_gxx():
    # the headers must be host-only C++14: g++, no hipcc, -std=c++14
    assert "-std=c++14" in open(os.path.join(EX, "Makefile")).read()
    _make()
    for p in PROGS:
        assert os.path.exists(os.path.join(EX, "_build", p))


def test_reference_include_paths_exist():
    inc = os.path.join(ROOT, "include", "uammd")
    for h in ["uammd.cuh", "Interactor/PairForces.cuh", "Interactor/NeighbourList/CellList.cuh",
              "Interactor/Potential/Potential.cuh", "Integrator/VerletNVT.cuh", "Integrator/BrownianDynamics.cuh",
              "Integrator/BDHI/BDHI_FCM.cuh", "Integrator/BDHI/BDHI_PSE.cuh", "Integrator/BDHI/BDHI_EulerMaruyama.cuh", "Integrator/BDHI/BDHI_Cholesky.cuh",
              "Interactor/NeighbourList/VerletList.cuh", "misc/LanczosAlgorithm.cuh", "Interactor/SpectralEwaldPoisson.cuh", "utils/checkpoint.h", "Integrator/Hydro/BDHI_quasi2D.cuh", "Integrator/BDHI/FIB.cuh", "Integrator/BDHI/FIB/FIB.cuh", "Integrator/Hydro/ICM.cuh",
              # round 6: more of the reference's include paths whose content the build has (src/<path>)
              "Integrator/BDHI/BDHI.cuh", "Integrator/BDHI/FCM/FCM_kernels.cuh", "Integrator/BDHI/FCM/utils.cuh", "Interactor/NBody.cuh", "Interactor/NBodyBase.cuh",
              "Interactor/Potential/ParameterHandler.cuh", "Interactor/NeighbourList/VerletList/NeighbourContainer.cuh", "misc/ParameterUpdatable.h", "System/Log.h",
              "misc/LanczosAlgorithm/MatrixDot.h", "utils/ForceEnergyVirial.cuh", "misc/ChevyshevUtils.cuh", "utils/execution_policy.cuh"]:
        assert os.path.exists(os.path.join(inc, h)), h
        assert os.path.exists(os.path.join("/root/reference/src", h)) or not os.path.isdir("/root/reference/src"), h + " is not a path of the reference"


def test_every_header_compiles_on_its_own():
    """Each file under include/uammd as the only include of a translation unit, through g++ -std=c++14 (the headers' contract: plain host
    C++; what needs device code is behind __HIPCC__) — in single precision, and with -DDOUBLE_PRECISION for the headers that do not
    refuse that build by design (the modules with a single-precision backend only stop with an #error that says so)."""
    from concurrent.futures import ThreadPoolExecutor
    inc = os.path.join(ROOT, "include", "uammd")
    headers = sorted(os.path.relpath(os.path.join(d, f), inc) for d, _, fs in os.walk(inc) for f in fs if f.endswith((".h", ".cuh", ".hpp")))
    assert len(headers) > 60

    def compile_one(job):
        h, dp = job
        if h.endswith(".hip.hpp") or h == "utils/ParticleSorter.cuh":
            return job, 0, ""           # device templates / device iterators: hipcc only (exercised by the programs under tests/cxx and examples)
        cmd = ["g++", "-std=c++14", "-fsyntax-only", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", inc, "-I", os.path.join(ROOT, "include")]
        r = subprocess.run(cmd + (["-DDOUBLE_PRECISION"] if dp else []) + ["-include", h, "/dev/null"], capture_output=True, text=True)
        return job, r.returncode, r.stderr

    with ThreadPoolExecutor(8) as pool:
        results = list(pool.map(compile_one, [(h, dp) for h in headers for dp in (False, True)]))
    refused = 0
    for (h, dp), rc, err in results:
        if dp and rc != 0 and "single-precision backend only" in err:
            refused += 1
            continue
        assert rc == 0, (h, "DOUBLE_PRECISION" if dp else "single precision", err[-1500:])
    assert 5 <= refused <= 25, refused


def test_header_has_no_oracle_or_cpu_fallback():
    src = open(os.path.join(ROOT, "include", "uammd", "uammd.h")).read()
    assert "oracle" not in src.lower()


@pytest.mark.gpu
@pytest.mark.parametrize("prog,args", [("custom_potential", []), ("ibm_library_mode", []), ("bd_readme", ["100000"]), ("lj_benchmark", ["131072", "50", "64"]),
                                       ("fcm_selfmobility", []), ("pse_selfmobility", []), ("poisson_two_charges", []), ("checkpoint", []), ("quasi2d_selfmobility", []), ("particle_group", []), ("particle_group", ["600", "7"]),
                                       ("custom_transverser", []), ("nbody", []), ("module_lifetime", []), ("particle_sorter", []), ("tabulated_function", []), ("container", []), ("chebyshev_grid", []), ("api_corners", []), ("dp_euler_maruyama", []), ("dp_bd", [])])
def test_examples_run(prog, args):
    _make()
    r = subprocess.run([os.path.join(EX, "_build", prog)] + args, capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prog,args", [("lj_slab", ["32768", "40"]), ("fcm_slab", ["64", "20000"]), ("fcm_slab", ["36", "4000"])])
def test_slab_drivers_world1(prog, args, tmp_path):
    """The C++14 multi-GPU drivers (include/uammd/Distributed.h over uammd::Comm = RCCL behind the C ABI) as ONE rank that is its own
    neighbour through the periodic z faces — every message of the N-rank schedule is sent and received, through RCCL, on the one GPU a
    test box has — compared inside the program with the single-domain classes of uammd.h.  (RCCL refuses two ranks on one device: runs
    with N > 1 processes need N GPUs; the process-level N > 1 runs of the same library kernels are tests/test_gpu_world2.py.)"""
    _make()
    r = subprocess.run([os.path.join(EX, "_build", prog), "0", "1", str(tmp_path / "comm.id")] + args, capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "world 1:" in r.stdout


@pytest.mark.gpu
def test_reference_benchmark_program_runs(tmp_path):
    """examples/_build/ref_benchmark = the REFERENCE's examples/misc/benchmark.cu, compiled from where it lies by plain g++ against
    include/uammd and linked with libuammd_hip (examples/Makefile; built in the container that holds the reference tree, the binary
    travels).  Its own defaults: 2^20 particles from initLattice(fcc) in a 128^3 box, VerletNVT::GronbechJensen, PairForces<LJ,
    VerletList> with rcutmult 1.2, 500 + 500 steps, "mean FPS" on the log."""
    exe = os.path.join(EX, "_build", "ref_benchmark")
    if not os.path.exists(exe):
        pytest.skip("ref_benchmark was not built (no reference tree where `make -C examples` ran)")
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    import re
    m = re.search(r"mean FPS: ([0-9.]+)", r.stdout + r.stderr)
    assert m, "the program did not report its rate"
    assert float(m.group(1)) > 200.0     # ~90 on the GTX 980 of the reference's comment; > 2000 measured on MI355X
    assert (tmp_path / "data.main.benchmark").exists()   # it wrote its default parameter file through InputFile's reader


def test_signal_semantics_host_only(tmp_path):
    """signal / connection / scoped_connection (the reference's nod::unsafe_signal, ParticleData.cuh:110-125) under AddressSanitizer, no
    device call: connect / emit / disconnect, a listener destroyed before the signal (round 4's use-after-free), slots that disconnect
    themselves or connect others during an emission, a connection that outlives its signal."""
    exe = str(tmp_path / "signal_semantics")
    rocm = "/opt/rocm"
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include",
                        "-I" + os.path.join(ROOT, "include", "uammd"), os.path.join(ROOT, "tests", "cxx", "signal_semantics.cpp"), "-o", exe,
                        "-L" + os.path.join(ROOT, "uammd_amd", "lib"), "-luammd_hip", f"-L{rocm}/lib", "-lamdhip64",
                        "-Wl,-rpath," + os.path.join(ROOT, "uammd_amd", "lib"), f"-Wl,-rpath,{rocm}/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "signal semantics ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prog,expect", [("ref_NeighbourListIterator", "mean FPS"), ("ref_neighbour_list", None), ("ref_signals", "No work needs to be done"),
                                         ("ref_temporary_memory", None), ("ref_execution_policy", "0 1 2")])
def test_more_reference_programs_run(prog, expect, tmp_path):
    """The reference's examples/advanced/{NeighbourListIterator,signals,temporary_memory,execution_policy}.cu and uammd_as_a_library/neighbour_list.cu,
    compiled from where they lie by hipcc against include/uammd (examples/Makefile, the documented user-side spellings replaced) and RUN:
    a kernel of the program's own over CellList::getNeighbourContainer() driving 500 steps of VerletNVT, BasicNeighbourListBase on a
    thrust vector with the list walked from a kernel, from thrust and downloaded, signal / connection objects, thrust vectors on
    System::allocator_thrust, thrust::sort under uammd::cached_device_execution_policy.on(stream)."""
    exe = os.path.join(EX, "_build", prog)
    if not os.path.exists(exe):
        pytest.skip(prog + " was not built (no reference tree where `make -C examples` ran)")
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    if expect:
        assert expect in r.stdout + r.stderr


# Every other program of the compile corpus (tests/test_reference_programs_compile.py), built in full by examples/Makefile and RUN: the
# thirteen tutorials of basic_concepts/, LJMultipleTypes, checkpoint, customPotentials, error_handling (which provokes the exceptions it
# shows how to handle), the FCM / BDHI / q2D integrator examples, the Poisson and ExternalForces examples, electrostatic_forces.
REF_EXAMPLES = ["1-system", "2-hello_world", "3-more_system", "4-uammd_types", "5-particle_data", "6-particle_data2", "7-moving_particles", "8-interacting_particles",
                "9-reading_parameters", "10-initial_configuration", "11-measuring_things", "12-your-first-integrator", "13-your-first-interactor", "LJMultipleTypes",
                "checkpoint", "customPotentials", "error_handling", "FCM", "BDHI", "q2D", "Poisson", "external", "electrostatic_forces"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_EXAMPLES)
def test_reference_example_programs_run(name, tmp_path):
    """Each program runs in a directory of its own to its end, or for 20 s where it hard-codes a long run (BDHI.cu), without an abort, an
    uncaught exception or a non-finite number in what it prints; q2D.cu is given the parameter file its main() asks for (both schemes)."""
    exe = os.path.join(EX, "_build", "ref_" + name)
    if not os.path.exists(exe):
        pytest.skip("ref_%s was not built (no reference tree where `make -C examples` ran)" % name)
    runs = [[]]
    if name == "q2D":
        runs = []
        for scheme in ("quasi2D", "true2D"):
            (tmp_path / ("data." + scheme)).write_text("boxSize 64 64\nnumberSteps 2000\nprintSteps 500\ndt 0.01\nrelaxSteps 0\nviscosity 1\ntemperature 1\n"
                                                      "hydrodynamicRadius 1\nscheme %s\nnumberParticles 4096\nloadParticles 0\noutput pos.%s\n" % (scheme, scheme))
            runs.append(["data." + scheme])
    for args in runs:
        with open(tmp_path / "out.txt", "w") as out, open(tmp_path / "err.txt", "w") as err:
            p = subprocess.Popen([exe] + args, cwd=tmp_path, stdout=out, stderr=err)
            try:
                p.wait(timeout=20)
                stopped = False
            except subprocess.TimeoutExpired:
                p.terminate()
                p.wait(timeout=30)
                stopped = True
        text, errors = open(tmp_path / "out.txt").read(), open(tmp_path / "err.txt").read()
        assert stopped or p.returncode == 0, (p.returncode, errors[-2000:])
        assert "terminate called" not in errors and "Aborted" not in errors, errors[-2000:]
        if name != "error_handling":
            assert "EXCEPTION" not in errors and "ERROR" not in errors, errors[-2000:]
        words = text.split()
        assert not any(w.lower() in ("nan", "-nan", "inf", "-inf") for w in words[:200000]), "a non-finite number in the output"
    if name == "q2D":
        for scheme in ("quasi2D", "true2D"):
            rows = [l.split() for l in open(tmp_path / ("pos." + scheme)) if not l.startswith("#")]
            a = np.array([[float(x) for x in r[:2]] for r in rows if len(r) >= 2])
            assert a.shape == (4 * 4096, 2) and np.isfinite(a).all()


@pytest.mark.gpu
def test_reference_parameter_updatable_program_steps(tmp_path):
    """examples/_build/ref_ParameterUpdatable = the reference's examples/advanced/ParameterUpdatable.cu compiled from where it lies (its
    cuda::std::plus as thrust::plus): VerletNVT::GronbechJensen over ExternalForces<MovingWall> — a functor that is ParameterUpdatable and
    reads the simulation time in its device code —, a small Interactor that overrides updateSimulationTime, and PairForces with a
    Potential of the program's own whose Transverser is rebuilt when the temperature or the time changes.  The program hard-codes 1e6
    steps: it is given a few seconds, must have written twelve frames by then (one per 1000 steps; t = 60), and is stopped.  (Twelve: while
    PairForces did not hand the parameter updates on to its Potential — ParameterUpdatableDelegate, PairForces.cuh:25 — the example's
    epsilon(t) stayed at its initial 0, the particles overlapped freely under the wall's pull, 0 * inf made a NaN and the cell list
    refused the positions after 1000 to 10000 steps; the same system through Potential::LJ runs on: tools/pu_physics_check.py.)"""
    import time
    exe = os.path.join(EX, "_build", "ref_ParameterUpdatable")
    if not os.path.exists(exe):
        pytest.skip("ref_ParameterUpdatable was not built (no reference tree where `make -C examples` ran)")
    with open(tmp_path / "out.txt", "w") as out, open(tmp_path / "err.txt", "w") as err:
        p = subprocess.Popen([exe], cwd=tmp_path, stdout=out, stderr=err)
        deadline = time.time() + 120
        while time.time() < deadline and p.poll() is None and os.path.getsize(tmp_path / "out.txt") < 14 * 16384 * 40:   # (a frame is ~0.6 MB)
            time.sleep(0.5)
        running = p.poll() is None
        if running:
            p.terminate()
        p.wait(timeout=30)
    text = open(tmp_path / "out.txt").read()
    errors = open(tmp_path / "err.txt").read()
    assert running or p.returncode == 0, errors[-3000:]
    frames = text.split("#Lx=")[1:]
    assert len(frames) >= 12, (len(frames), errors[-2000:])
    rows = np.array([[float(x) for x in line.split()[:3]] for line in frames[10].splitlines()[1:16385]])
    assert rows.shape == (16384, 3) and np.isfinite(rows).all()
    assert np.abs(rows[:, 0]).max() <= 16.0 + 1e-3 and np.abs(rows[:, 2]).max() <= 20.0 + 1e-3      # inside the box it prints (apply_pbc)
    assert "ERROR" not in errors and "EXCEPTION" not in errors, errors[-3000:]


@pytest.mark.gpu
def test_reference_LJ_program_runs(tmp_path):
    """examples/_build/ref_LJ = the REFERENCE's examples/misc/LJ.cu compiled from where it lies by hipcc against include/uammd
    (PairForces<Potential::LJ> on the library's fused path + two ExternalForces<HarmonicWall> — a device functor of the program's own —
    on ParticleGroups selected by type, VerletNVT::GronbechJensen, InputFile).  Run on its own parameter file format; the walls pull the
    two species to z = -Lz/4 and +Lz/4."""
    exe = os.path.join(EX, "_build", "ref_LJ")
    if not os.path.exists(exe):
        pytest.skip("ref_LJ was not built (no reference tree where `make -C examples` ran)")
    # (a dilute mixture — the program's default density with sigma = 2 for one species is close packed and does not move in a short run)
    n = 4096
    (tmp_path / "data.main.lj").write_text(f"boxSize 60 60 60\nnumberParticles {n}\ndt 0.005\nnumberSteps 6000\nprintSteps 5999\n"
                                           f"outputFile {tmp_path / 'out.dat'}\ntemperature 0.5\nfriction 1.0\n")
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [l.split() for l in open(tmp_path / "out.dat") if l.strip() and not l.startswith("#")]
    assert len(rows) >= n
    import numpy as np
    last = np.array(rows[-n:], dtype=float)
    assert np.isfinite(last).all()
    # columns: x y z radius type.  The harmonic walls (k = 0.1, ExternalForces on the ParticleGroups of type 0 and type 1) pull type 0
    # towards z = -Lz/4 = -15 and type 1 towards + 15.  The program's cross interaction is 8 kT deep (epsilon 4 at T = 0.5), so the species
    # stick to each other and only partly demix in 30 time units: the test asks for the right DIRECTIONS (the exact wall force on a
    # group, and several arrays through a tuple, are checked to the bit in examples/custom_potential.hip)
    z, kind = last[:, 2], last[:, 4]
    z0, z1 = z[kind == 0].mean(), z[kind == 1].mean()
    print("mean z of type 0:", z0, " of type 1:", z1)
    assert (kind == 0).sum() > n // 4 and (kind == 1).sum() > n // 4
    assert z0 < -1.0 and z1 > 1.0


@pytest.mark.gpu
def test_reference_acceptance_programs_run(tmp_path):
    """examples/_build/ref_test_{FCM,PSE} = the REFERENCE's own acceptance programs test/BDHI/FCM/FCM.cu and test/BDHI/PSE/PSE.cu (plain
    programs; test.bash beside them drives them and plots), compiled from where they lie by hipcc against include/uammd and linked with
    libuammd_hip.  Their user-side Interactor pulls particle 0 through `pg->getNumberParticles()` and the CPU property accessors — the
    base class's `pg` must be the group of all particles for a module built from a ParticleData, as in the reference (it was null until
    round 5: the programs crashed).  Self mobility in cubic boxes of 8 .. 128 hydrodynamic radii (FCM) and over psi and L (PSE): every
    entry of |1 - M / M0| within the criterion of the reference's script, 2 / (L / rh)^6 + tolerance."""
    tol = 1e-3
    fcm = os.path.join(EX, "_build", "ref_test_FCM")
    pse = os.path.join(EX, "_build", "ref_test_PSE")
    if not (os.path.exists(fcm) and os.path.exists(pse)):
        pytest.skip("the acceptance programs were not built (no reference tree where `make -C examples` ran)")
    r = subprocess.run([fcm, "selfMobilityCubicBox", "0", "1", "1", str(tol)], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [[float(x) for x in l.split()] for l in open(tmp_path / "selfMobilityCubicBox.test") if l.strip()]
    assert len(rows) == 20
    for row in rows:
        bar = 2.0 / row[0] ** 6 + tol
        assert max(row[1:]) <= bar, (row[0], max(row[1:]), bar)
    print("FCM self mobility: largest |1 - M/M0| %.2e over 20 boxes" % max(max(row[1:]) for row in rows))
    r = subprocess.run([pse, "selfMobility", "0", "1", "1", str(tol)], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("selfMobility_pullForce"))
    assert len(files) >= 3
    worst = 0.0
    for f in files:
        for l in open(tmp_path / f):
            if l.startswith("#") or not l.strip():
                continue
            L, dev = (float(x) for x in l.split()[:2])
            worst = max(worst, abs(dev))
            assert abs(dev) <= 2.0 / L ** 6 + tol, (f, L, dev)
    print("PSE self mobility: largest deviation %.2e over %d values of psi" % (worst, len(files)))
    # pair mobility against the open-boundary formula, boxes from 2.1 distances (the kernel's support exceeds the first grids: the reference
    # says so and goes on, BDHI_FCM.cuh:58-64) to 100 radii: the periodic images' share shrinks with the box
    r = subprocess.run([fcm, "pairMobilityCubicBox", "0", "1", "1", str(tol)], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Kernel support is too big" in r.stderr
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("pairMobilityCubicBox.dist"))
    assert len(files) == 3
    for f in files:
        rows = [[float(x) for x in l.split()] for l in open(tmp_path / f) if l.strip() and not l.startswith("#")]
        assert len(rows) == 20 and all(np.isfinite(row).all() for row in rows)
        # (diagonal entries: columns 1, 5, 9)
        first, last = max(rows[0][1], rows[0][5], rows[0][9]), max(rows[-1][1], rows[-1][5], rows[-1][9])
        assert last < 0.1 * first and last < 0.05, (f, first, last)
    # test/BDHI/FIB/FIB.cu, same sweep on the staggered grid: its three-point Peskin kernel is translation invariant to about a percent
    fib = os.path.join(EX, "_build", "ref_test_FIB")
    if os.path.exists(fib):
        os.remove(tmp_path / "selfMobilityCubicBox.test")
        r = subprocess.run([fib, "selfMobilityCubicBox", "0", "1", "1", str(tol)], cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        rows = [[float(x) for x in l.split()] for l in open(tmp_path / "selfMobilityCubicBox.test") if l.strip()]
        assert len(rows) >= 20 and max(max(row[1:]) for row in rows) <= 3e-2
        print("FIB self mobility: largest |1 - M/M0| %.2e over %d boxes" % (max(max(row[1:]) for row in rows), len(rows)))


@pytest.mark.gpu
def test_reference_acceptance_programs_double_precision(tmp_path):
    """The reference's acceptance programs built as THEIR OWN Makefiles build them, with -DDOUBLE_PRECISION (test/BDHI/FCM/Makefile:9,
    test/BDHI/quasi2D/Makefile:2), and run as their scripts run them:
    * test/BDHI/FCM/FCM.cu selfMobilityCubicBox at the script's tolerance of 1e-14 (test.bash:33,44-58) — twenty boxes of 8 .. 128
      hydrodynamic radii, the first of them smaller than the spreading kernel (support 27 on a 24^3 grid: the reference prints its ERROR
      line and goes on, and so does the double-precision build) — judged by the script's own criterion: from the fifth row on every
      |1 - M / M0| <= 2 / (L1 / rh)^6 + tolerance with L1 the first row's box;
    * test/BDHI/quasi2D/q2D.cu selfMobility for both hydrodynamic kernels as checkSelfMobility writes its input (test.bash:98-125; the script
      plots |1 - M / M0| against L / a and has no number to pass): the deviation falls with the box and is below 5e-3 at L = 64 a,
      1.5e-3 at 128 a, 1e-3 at 256 a."""
    fcm = os.path.join(EX, "_build", "ref_test_dp_FCM")
    q2d = os.path.join(EX, "_build", "ref_test_dp_q2D")
    if not (os.path.exists(fcm) and os.path.exists(q2d)):
        pytest.skip("the double-precision acceptance programs were not built (no reference tree where `make -C examples` ran)")
    tol = 1e-14
    r = subprocess.run([fcm, "selfMobilityCubicBox", "0", "1", "1", str(tol), "0"], cwd=tmp_path, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Kernel support is too big" in r.stderr
    rows = [[float(x) for x in l.split()] for l in open(tmp_path / "selfMobilityCubicBox.test") if l.strip()]
    assert len(rows) == 20
    bar = 2.0 / rows[0][0] ** 6 + tol
    worst = max(abs(v) for row in rows[4:] for v in row[1:])
    print("FCM, double precision, tolerance 1e-14: largest |1 - M/M0| from the fifth box on %.3e, the script's bar %.3e" % (worst, bar))
    assert worst <= bar
    a, vis = 1.3, 1.7
    for scheme in ("quasi2D", "true2D"):
        devs = []
        for l in (32, 64, 128, 256):
            M0 = 1.0 / (6 * np.pi * vis * a) / (1 + 4.41 / l) if scheme == "quasi2D" else (np.log(l) - 1.3105329259115095183) / (4 * np.pi * vis)
            (tmp_path / "in.q2d").write_text(f"scheme {scheme}\ntest selfMobility\nboxSize {l * a:.15g} {l * a:.15g}\ncells -1 -1\ndt 1\nviscosity {vis}\n"
                                             f"temperature 0\nhydrodynamicRadius {a}\ntolerance 1e-6\nF {a / M0:.15g}\noutput /dev/stdout\n")
            r = subprocess.run([q2d, "in.q2d"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            vals = [float(l.split()[0]) for l in r.stdout.splitlines() if l.strip()]   # (a line is the real3 M: the script reads column 1)
            assert vals, r.stdout[-500:] + r.stderr[-500:]
            devs.append(max(abs(1.0 - v / M0) for v in vals))
        print("q2D %s, double precision: |1 - M/M0| at L/a = 32, 64, 128, 256: %s" % (scheme, " ".join("%.2e" % d for d in devs)))
        assert devs[1] <= 5e-3 and devs[2] <= 1.5e-3 and devs[3] <= 1e-3 and devs[3] <= devs[0]   # (measured 2.7e-3 / 6.5e-4 / 7.7e-5 and 1.1e-3 / 7.8e-4 / 6.2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Lanczos", "Cholesky"])
def test_reference_rpy_acceptance_pipeline_double_precision(mode, tmp_path):
    """test/BDHI/Lanczos_Cholesky as its test.bash runs it, built as its Makefile builds it (-DDOUBLE_PRECISION, Makefile:3): BDHI.cu on 5000
    spheres of two radii, the big one pulled, its positions piped into the reference's checker process.cpp — and judged by the script's OWN
    criterion for both modes (test.bash:24-47): the largest deviation of f(r) and of g(r) from the Rotne-Prager-Yamakawa formulas for
    unequal spheres <= 1e-7.  (EulerMaruyama<BDHI::Lanczos> takes the matrix-free product uammd_rpy_nbody_mdot_f64, EulerMaruyama<BDHI::
    Cholesky> the dense matrix with rocBLAS dsymv.)  Ten steps instead of the script's hundred; the seed named so that a run is
    repeatable (BDHI.cu:16,127 seeds from the clock otherwise)."""
    prog = os.path.join(EX, "_build", "ref_test_dp_BDHI")
    proc = os.path.join(EX, "_build", "ref_process_bdhi")
    if not (os.path.exists(prog) and os.path.exists(proc)):
        pytest.skip("the acceptance program was not built (no reference tree where `make -C examples` ran)")
    (tmp_path / "data.main").write_text("N 5000\nboxSize 4 4 4\nradius_min 0.38173\nradius_max 1.89538\noutfile /dev/stdout\ntemperature 0\n"
                                        "viscosity 1.2131\ndt 10\ntolerance 1e-8\nnsteps 10\nprintSteps 1\nmode %s\nseed 20260930\n" % mode)
    run = subprocess.run([prog], cwd=tmp_path, capture_output=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    chk = subprocess.run([proc], cwd=tmp_path, input=run.stdout, capture_output=True, timeout=600)
    assert chk.returncode == 0, chk.stderr[-2000:]
    d = np.array([[float(x) for x in l.split()[:3]] for l in chk.stdout.decode().splitlines() if l.strip() and not l.startswith("#")])
    assert d.shape[0] > 40000
    print("RPY acceptance, double precision, %s: %d pairs; largest deviation of f %.2e, of g %.2e (the script's bar: 1e-7 each)" %
          (mode, d.shape[0], d[:, 1].max(), d[:, 2].max()))
    assert d[:, 1].max() <= 1e-7 and d[:, 2].max() <= 1e-7


@pytest.mark.gpu
def test_reference_rpy_acceptance_pipeline(tmp_path):
    """test/BDHI/Lanczos_Cholesky of the reference, as its test.bash runs it: BDHI.cu (EulerMaruyama<BDHI::Lanczos> on 5000 spheres of two
    radii, the big one pulled; a user Interactor writing through the CPU accessors and getIdOrderedIndices) built against include/uammd,
    its positions piped into the checker the reference ships beside it (process.cpp, plain g++): every pair's f(r) and g(r) against the
    Rotne-Prager-Yamakawa formulas for unequal spheres.  The script's bar (1e-7) is for its DOUBLE_PRECISION build; the headers' real is
    float: f to 1e-5; g 2e-6 in the median and, beyond contact (where it is not about to vanish), within 1e-2 for 99.9 % of the pairs
    (the checker extracts g from the displacement's component along r: ill conditioned for the pairs that lie across the pull — the
    worst pair of a run sits between 3e-3 and 2e-2 depending on the positions).  The program seeds its positions from the clock unless its
    data file names a seed (BDHI.cu:16,127): the test names one, so that a run is a repeatable statement."""
    seed = int(os.environ.get("UAMMD_RPY_ACCEPTANCE_SEED", "20260930"))
    prog = os.path.join(EX, "_build", "ref_test_BDHI")
    proc = os.path.join(EX, "_build", "ref_process_bdhi")
    if not (os.path.exists(prog) and os.path.exists(proc)):
        pytest.skip("the acceptance program was not built (no reference tree where `make -C examples` ran)")
    (tmp_path / "data.main").write_text("N 5000\nboxSize 4 4 4\nradius_min 0.38173\nradius_max 1.89538\noutfile /dev/stdout\ntemperature 0\n"
                                        "viscosity 1.2131\ndt 10\ntolerance 1e-8\nnsteps 10\nprintSteps 1\nmode Lanczos\nseed %d\n" % seed)
    run = subprocess.run([prog], cwd=tmp_path, capture_output=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    chk = subprocess.run([proc], cwd=tmp_path, input=run.stdout, capture_output=True, timeout=600)
    assert chk.returncode == 0, chk.stderr[-2000:]
    d = np.array([[float(x) for x in l.split()[:3]] for l in chk.stdout.decode().splitlines() if l.strip() and not l.startswith("#")])
    assert d.shape[0] > 80000
    apart = d[:, 0] > 0.38173 + 1.89538
    print("RPY acceptance: %d pairs; f deviation max %.2e; g deviation median %.2e, beyond contact 99.9 %% within %.2e, max %.2e" %
          (d.shape[0], d[:, 1].max(), np.median(d[:, 2]), np.quantile(d[apart, 2], 0.999), d[apart, 2].max()))
    assert d[:, 1].max() <= 1e-5
    assert np.median(d[:, 2]) <= 1e-4 and np.quantile(d[apart, 2], 0.999) <= 1e-2 and d[apart, 2].max() <= 0.2


@pytest.mark.gpu
def test_user_side_saru_matches_the_golden_streams():
    """third_party/saruprng.cuh in a user kernel and on the host (examples/_build/saru_user = tests/cxx/saru_user.hip): the integer streams
    of the one-, two- and three-seed constructors against tests/golden/saru_u32.npz for every seed triple of the file, device == host, and
    the Gaussian draws' moments."""
    _make()
    exe = os.path.join(EX, "_build", "saru_user")
    g = np.load(os.path.join(ROOT, "tests", "golden", "saru_u32.npz"))
    for k, s in enumerate(g["seeds"]):
        r = subprocess.run([exe] + [str(int(x)) for x in s], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l.strip()}
        for name, key in (("u32_1", "one"), ("u32_2", "two"), ("u32_3", "three")):
            assert np.array_equal(np.array([int(x) for x in lines[name]], dtype=np.uint32), g[key][k]), (name, s)
    mean, second = (float(x) for x in lines["moments"])
    assert abs(mean) <= 0.06 and abs(second - 1.0) <= 0.08      # 3000 draws of N(0, 1)


# The reference's own GoogleTest programs of the starred rows (SURVEY 8c's pins), built by examples/Makefile from where they lie with
# -DDOUBLE_PRECISION -DMAXLOGLEVEL=1 (test/CMakeLists.txt:4,9) against include/uammd + tests/cxx/gtest_lite; every TEST at its own tolerance:
#   utils/ParticleSorter.cu (2)         misc/ibm/test_ibm_regular.cu (7: constant kernel counts, Peskin 1e-10, adjoint 1e-4)
#   misc/ibm/test_ibm.cu (5: IBM<Kernel, Grid> with the USER's grid, support and quadrature: Gaussian with a per-particle support on
#   chebyshev::doublyperiodic::Grid (misc/ChevyshevUtils.cuh), complex quantity, every node of 16^3 / 128^3 against the analytic window
#   incl. which nodes are exactly zero; spread then gather with Clenshaw-Curtis weights against the closed form, 1e-11)
#   misc/lanczos/test_lanczos.cu (7: identity .. dense SPD up to 511 x 511, 1e-7)   BDHI/FCM/fcm_test.cu (2: Hasimoto 1e-8 at 288^3)
#   BDHI/PSE/pse_test.cu (4: Hasimoto 1e-8, self diffusion 1e-2)
# and, for the other consumers of the spread / FFT / gather engine (their double-precision builds are csrc/f64.hip):
#   BDHI/quasi2D/quasi2d_test.cu (5: self mobility of True2D / Quasi2D at seven box sizes 1e-3, fluctuation-dissipation 1e-2)
#   Potentials/Poisson/TriplyPeriodic/test_poisson.cu (2: three charges against the analytic field 1e-3; L -> infinity extrapolation over
#   109 box sizes x 6 distances 1e-4)      .../test_tp_quadrupole.cu (1: point quadrupole, tolerance 1e-14, window support 41: field 1e-8)
REF_GTESTS = {"ParticleSorter": 2, "test_ibm_regular": 7, "test_ibm": 5, "test_lanczos": 7, "fcm_test": 2, "pse_test": 4, "quasi2d_test": 5, "test_poisson": 2,
              "test_tp_quadrupole": 1}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(REF_GTESTS))
def test_reference_unit_tests_run(name, tmp_path):
    import re
    exe = os.path.join(EX, "_build", "ref_gtest_" + name)
    if not os.path.exists(exe):
        pytest.skip("ref_gtest_%s was not built (no reference tree where `make -C examples` ran)" % name)
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=1200)
    out = r.stdout + r.stderr
    print(out[-6000:])
    ran = re.search(r"\[==========\] (\d+) tests ran", out)
    assert ran and int(ran.group(1)) == REF_GTESTS[name], "not every TEST of the file ran"
    failed = re.findall(r"^\[  FAILED  \] (\S+)$", out, flags=re.M)
    # Statistical TESTs, seeded from the clock as in the reference (System.h:90-96; std::random_device in quasi2d_test.cu): pse_test's
    # <dx^2> over 1000 draws against 2 T M0 with an absolute bar of 1e-2 is a 2.3-sigma criterion per component (sigma = 0.0955
    # sqrt(2 / 1000); measured: 5 of 65 runs of one TEST fail, 6 % expected for three components); quasi2d_test's 50000 one-step variances
    # against 1 % are a 1.6-sigma criterion per component (sqrt(2 / 50000) = 0.63 %: 11 % per component, 21 % per TEST, 38 % per run of the
    # file; seen: 4 failing runs of 10.  With 800 000 samples the same quantity sits at +0.19 / -0.04 / -0.19 / +0.02 % of 2 T dt M for the
    # four components, sigma 0.16 %: tools/q2d_fdt_stats.cpp — no bias).  A correct sampler fails such a TEST now and then — with
    # GoogleTest in the reference too: up to five repetitions of exactly those TESTs (six tries of a 21 % event: 1e-4).
    repeats = 5
    while failed and repeats > 0 and all("SelfDiffusion" in f or "FluctuationDissipation" in f for f in set(failed)):
        repeats -= 1
        r = subprocess.run([exe, "--gtest_filter=" + ":".join(sorted(set(failed)))], cwd=tmp_path, capture_output=True, text=True, timeout=1200)
        out = r.stdout + r.stderr
        print(out[-3000:])
        failed = re.findall(r"^\[  FAILED  \] (\S+)$", out, flags=re.M)
    assert r.returncode == 0 and not failed, failed


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["EulerMaruyama", "MidPoint", "AdamsBashforth", "Leimkuhler"])
def test_reference_bd_program_free_diffusion(scheme, tmp_path):
    """examples/_build/ref_test_BD = the reference's test/BD/BD.cu (compiled from where it lies), run as its test.bash runs it — free
    particles, dt = 1, viscosity 1 / (6 pi), a = 1, T = 1: D0 = 1 — and judged by that script's criterion: the slope of the mean square
    displacement per coordinate against 2 D0 dt (the script fits the first five lags of the msd tool's output; here the frames are read
    directly).  The script loops over AdamsBashforth, MidPoint and EulerMaruyama; Leimkuhler, whose noise is correlated over one step,
    is judged on lags 5..15."""
    exe = os.path.join(EX, "_build", "ref_test_BD")
    if not os.path.exists(exe):
        pytest.skip("ref_test_BD was not built (no reference tree where `make -C examples` ran)")
    # (Leimkuhler is judged on lags 5 .. 15: windows that long are few and overlap in a short run — with 4096 particles and 24 frames
    # the fitted slope scattered by ~5 %, the bar itself, and the test failed every other run; 16384 particles and 64 frames: ~1 %)
    n, steps = (16384, 64) if scheme == "Leimkuhler" else (4096, 24)
    (tmp_path / "data.main").write_text(f"""scheme {scheme}
potential none
boxSize 32 32 32
numberSteps {steps}
printSteps 1
relaxSteps -1
dt         1
numberParticles {n}
temperature    1
viscosity   {1 / (6 * np.pi):.14g}
hydrodynamicRadius 1
outfile {tmp_path / 'pos.dat'}
""")
    r = subprocess.run([exe, "data.main"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    frames = np.loadtxt(tmp_path / "pos.dat", comments="#", usecols=(0, 1, 2)).reshape(steps, n, 3)
    lags = range(5, 16) if scheme == "Leimkuhler" else range(1, 6)
    msd = np.array([((frames[lag:] - frames[:-lag]) ** 2).mean((0, 1)) for lag in lags])
    t = np.array(list(lags), float)
    slope = np.array([np.polyfit(t, msd[:, c], 1)[0] for c in range(3)])
    print(scheme, "msd slope / (2 D0 dt):", slope / 2.0)
    assert np.all(np.abs(slope / 2.0 - 1) < 0.05), slope
