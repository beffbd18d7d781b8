"""CPU test of the Mersenne-twister jump-ahead machinery (climt_amd/csrc/rrtmg_mt_jump.cpp): the characteristic polynomial found
by Berlekamp-Massey has degree 19937, and the windows formed from the segments' polynomials equal the sequential stream's --
for an unsharded call (first segment at draw 0) and for a shard that starts in the middle of the stream, with every run cut
into pieces.  (The device applies the same lists: tests/test_gpu_parity.py, against the sequential host stream.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jump_polynomials_reproduce_the_sequential_stream(tmp_path):
    exe = str(tmp_path / "mt_jump_selftest")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DRRTMG_MT_JUMP_SELFTEST", os.path.join(ROOT, "climt_amd", "csrc", "rrtmg_mt_jump.cpp"),
                           "-o", exe, "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "degree 19937" in r.stdout and "MISMATCH" not in r.stdout
    assert r.stdout.count(": ok") == 23 and "seed window" in r.stdout
