"""NUMA placement of the library's own host threads (csrc/numa_affinity.cpp), on the CPU with a mock sysfs tree.
dxtlt_transform_sharded starts one worker thread per shard; each binds itself to the CPUs local to its device's PCI
function (eight shards must not cross sockets blindly)."""
import ctypes as C
import os
import threading

import pytest


@pytest.fixture()
def lib(pkg):
    l = pkg.load()
    l.dxtlt_pci_local_cpulist.argtypes, l.dxtlt_pci_local_cpulist.restype = [C.c_char_p, C.c_char_p, C.c_size_t], C.c_int32
    l.dxtlt_bind_thread_to_cpulist.argtypes, l.dxtlt_bind_thread_to_cpulist.restype = [C.c_char_p], C.c_int32
    l.dxtlt_device_local_cpulist.argtypes, l.dxtlt_device_local_cpulist.restype = [C.c_int32, C.c_char_p, C.c_size_t], C.c_int32
    return l


def _mock_sysfs(tmp_path, bdf, local_cpulist=None, numa_node=None, node_cpulist=None):
    d = tmp_path / "bus" / "pci" / "devices" / bdf
    d.mkdir(parents=True)
    if local_cpulist is not None:
        (d / "local_cpulist").write_text(local_cpulist + "\n")
    if numa_node is not None:
        (d / "numa_node").write_text(f"{numa_node}\n")
    if node_cpulist is not None:
        n = tmp_path / "devices" / "system" / "node" / f"node{numa_node}"
        n.mkdir(parents=True)
        (n / "cpulist").write_text(node_cpulist + "\n")
    return str(tmp_path)


def _cpulist(lib, bdf):
    out = C.create_string_buffer(256)
    n = lib.dxtlt_pci_local_cpulist(bdf.encode(), out, 256)
    return n, out.value.decode()


def test_local_cpulist_comes_from_the_device_node(lib, tmp_path, monkeypatch):
    monkeypatch.setenv("DXTLT_SYSFS_ROOT", _mock_sysfs(tmp_path, "0000:c5:00.0", local_cpulist="64-127,192-255"))
    assert _cpulist(lib, "0000:c5:00.0") == (14, "64-127,192-255")
    assert _cpulist(lib, "0000:C5:00.0")[1] == "64-127,192-255"       # HIP prints bus ids in upper or lower case
    assert _cpulist(lib, "0000:05:00.0") == (0, "")                    # a function sysfs does not list


def test_numa_node_fallback_and_unknown_node(lib, tmp_path, monkeypatch):
    root = _mock_sysfs(tmp_path, "0000:05:00.0", numa_node=1, node_cpulist="8-15")
    monkeypatch.setenv("DXTLT_SYSFS_ROOT", root)
    assert _cpulist(lib, "0000:05:00.0") == (4, "8-15")
    root2 = _mock_sysfs(tmp_path / "b", "0000:06:00.0", numa_node=-1)
    monkeypatch.setenv("DXTLT_SYSFS_ROOT", root2)
    assert _cpulist(lib, "0000:06:00.0") == (0, "")                    # single-node host: the kernel says -1


def test_malformed_lists_are_refused(lib, tmp_path, monkeypatch):
    monkeypatch.setenv("DXTLT_SYSFS_ROOT", _mock_sysfs(tmp_path, "0000:07:00.0", local_cpulist="3-1,x"))
    assert _cpulist(lib, "0000:07:00.0") == (0, "")
    for bad in (b"", b"a-b", b"5-2", b"1,,2", b"99999"):
        assert lib.dxtlt_bind_thread_to_cpulist(bad) == 0
    assert lib.dxtlt_bind_thread_to_cpulist(None) == 0


def test_binding_moves_only_the_calling_thread(lib):
    """A worker thread binds itself; the thread that called into the library keeps its mask."""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("one CPU")
    target = allowed[-1]
    seen = {}

    def worker():
        seen["bound"] = lib.dxtlt_bind_thread_to_cpulist(f"{target}".encode())
        seen["mask"] = os.sched_getaffinity(threading.get_native_id())

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen["bound"] == 1 and seen["mask"] == {target}
    assert sorted(os.sched_getaffinity(0)) == allowed


def test_cpus_outside_the_process_mask_are_dropped(lib):
    allowed = sorted(os.sched_getaffinity(0))
    seen = {}

    def worker():
        # the node lists a CPU this process may not use (a container's cpuset): bind to the part that is ours
        seen["bound"] = lib.dxtlt_bind_thread_to_cpulist(f"{allowed[0]},1023".encode())
        seen["mask"] = os.sched_getaffinity(threading.get_native_id())
        seen["none"] = lib.dxtlt_bind_thread_to_cpulist(b"1023")

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    if 1023 in allowed:
        pytest.skip("a 1024-CPU host")
    assert seen["bound"] == 1 and seen["mask"] == {allowed[0]} and seen["none"] == 0


def test_device_lookup_without_a_device_is_empty(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = C.create_string_buffer(64)
    assert lib.dxtlt_device_local_cpulist(0, out, 64) == 0 and out.value == b""


@pytest.mark.gpu
def test_sharded_call_reports_its_shards_and_binds_its_workers(pkg, lib):
    """dxtlt_transform_sharded over four shards on the visible device(s): the stats name every shard once, the ranges tile
    the array, and -- when sysfs knows the device's node and this process may use some of its CPUs -- the workers were bound."""
    import numpy as np

    class Stat(C.Structure):
        _fields_ = [("device", C.c_int32), ("cpus_bound", C.c_int32), ("first_block", C.c_uint64), ("blocks", C.c_uint64),
                    ("seconds", C.c_double)]

    lib.dxtlt_sharded_last_stats.argtypes, lib.dxtlt_sharded_last_stats.restype = [C.POINTER(Stat), C.c_int32], C.c_int32
    mask_before = sorted(os.sched_getaffinity(0))
    n = 4 * 2048 * 8 + 24
    x = np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8)
    y = np.zeros_like(x)
    pkg.transform_sharded("bc1", False, x, y, pkg.Bc1TransformSettings(), 4)
    stats = (Stat * 8)()
    k = lib.dxtlt_sharded_last_stats(stats, 8)
    assert k == 4
    at = 0
    for s in stats[:k]:
        assert s.first_block == at and s.seconds > 0 and 0 <= s.device < pkg.load().dxtlt_device_count()
        at += s.blocks
    assert at == n // 8
    out = C.create_string_buffer(4096)
    if lib.dxtlt_device_local_cpulist(0, out, 4096) > 0:
        want = set()
        for part in out.value.decode().split(","):
            a, _, b = part.partition("-")
            want |= set(range(int(a), int(b or a) + 1))
        if want & os.sched_getaffinity(0):
            assert all(s.cpus_bound > 0 for s in stats[:k])
    assert sorted(os.sched_getaffinity(0)) == mask_before   # the caller's own mask is untouched
