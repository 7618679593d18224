"""The C ABI used from C: tests/c_abi_smoke.c is compiled against include/skred_amd.h, linked with
libskred_amd.so and run (bank mode: create / tables / upload / update / defer / run_queue / render_host /
download / error codes).  The compile-and-link half runs everywhere; running it needs the GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build(tmp_path, name="c_abi_smoke", hip=False):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O1", "-Wall", "-Werror", "-std=gnu11", "-I" + os.path.join(ROOT, "include"),
           os.path.join(HERE, name + ".c"), "-o", exe, "-L" + os.path.join(ROOT, "skred_amd"), "-lskred_amd", "-lm", "-lpthread",
           "-Wl,-rpath," + os.path.join(ROOT, "skred_amd")]
    if hip:      # the host looks at device buffers itself
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cmd += ["-I" + os.path.join(rocm, "include"), "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return exe


def test_c_host_compiles_and_links_against_the_headers(tmp_path):
    build(tmp_path)


@pytest.mark.gpu
def test_c_host_runs(tmp_path):
    out = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-1500:])
    assert out.stdout.startswith("OK")


def test_c_shard_host_compiles_and_links(tmp_path):
    build(tmp_path, "c_shard_smoke", hip=True)


@pytest.mark.gpu
def test_c_shard_host_runs(tmp_path):
    """skred_shard_* from C on the one GPU of the box: no collective, a host-supplied reduce, the library's RCCL."""
    out = subprocess.run([build(tmp_path, "c_shard_smoke", hip=True)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-1500:])
    assert out.stdout.strip().splitlines()[-1].startswith("OK")


def test_one_process_one_thread_per_rank(tmp_path):
    """tests/c_shard_threads.c: the single-process multi-GPU host form INTEGRATION.md describes (one thread per rank, each with its
    own skred_shard_t), rehearsed without devices -- host-memory steps, a barrier-based reduce in rank order -- for 1..8 ranks,
    serial and pipelined sequences, all threads in the library at once; bit-equal to the rank-ordered sum, error texts per
    thread.  No GPU needed: the custom steps never touch one."""
    out = subprocess.run([build(tmp_path, "c_shard_threads")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-1500:])
    assert out.stdout.startswith("OK")
