"""The fused Runge-Kutta step kernel, executed on the CPU: tests/cpu_emu/emu_fused.cpp runs the SAME phase functions
the HIP kernel runs (waves.jl_amd/csrc/fused_body.h + the host plan fused_plan.h) in host loops, under
AddressSanitizer/UBSan, and checks every field bit for bit against the C oracle.  This is how the tile / halo / one-sided
boundary / cylinder-culling index logic is verified without a GPU (GPU sanitizers are unavailable on the pool)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "cpu_emu")


def test_fused_kernel_body_bit_exact_under_asan():
    r = subprocess.run(["make", "-C", EMU, "emu_fused"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="4")
    r = subprocess.run([os.path.join(EMU, "emu_fused"), "quick"], capture_output=True, text=True, env=env, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS" in r.stdout and "MISMATCH" not in r.stdout
    assert r.stdout.count("bit-exact") >= 10
    # round 3: the JOB protocol of the resident launch (fused_body.h "jobs"), every block a thread running the device's own
    # protocol functions: a sequence of jobs whose tile -> block mapping changes, a launch that ends with a job, a launch that
    # leaves on its idle limit and is started again, and a forced give-up that must drain and leave the initial condition intact
    assert r.stdout.count("bit-exact (jobs)") == 3
    assert "launch drained, initial condition intact" in r.stdout
