"""The hand-over of the RCCL id in np_comm_init (rank 0 -> peers, over a loopback TCP socket or a file), run as
real separate processes without a GPU through np_comm_debug_exchange: world 2 and 4, both endpoint forms, peers
that start BEFORE rank 0 is listening, and the time-out when rank 0 never shows up."""
import ctypes as C
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent("""
    import ctypes as C, sys, time
    sys.path.insert(0, %r)
    from numpower_amd._lib import load
    rank, world, endpoint, delay, timeout = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], float(sys.argv[4]), float(sys.argv[5])
    lib = load()
    time.sleep(delay)
    buf = (C.c_ubyte * 128)(*([(7 * i + 3) %% 251 for i in range(128)] if rank == 0 else [0] * 128))
    rc = lib.np_comm_debug_exchange(rank, world, endpoint.encode(), buf, timeout)
    if rc != 0:
        print("ERR", lib.np_last_error().decode()); sys.exit(3)
    print("OK", bytes(buf).hex())
""") % str(ROOT)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(world, endpoint, rank0_delay=0.0, skip_rank0=False, timeout=20.0):
    procs = []
    for r in range(world):
        if r == 0 and skip_rank0:
            continue
        delay = rank0_delay if r == 0 else 0.0
        procs.append((r, subprocess.Popen([sys.executable, "-c", WORKER, str(r), str(world), endpoint, str(delay), str(timeout)],
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    out = {}
    for r, p in procs:
        so, se = p.communicate(timeout=90)
        out[r] = (p.returncode, so.strip(), se[-500:])
    return out


EXPECT = bytes((7 * i + 3) % 251 for i in range(128)).hex()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("form", ["tcp", "file"])
def test_every_peer_receives_rank0s_bytes(world, form, tmp_path):
    endpoint = "tcp://127.0.0.1:%d" % free_port() if form == "tcp" else str(tmp_path / "id")
    out = run_world(world, endpoint, rank0_delay=0.6)          # peers are up first and have to retry / poll
    for r in range(world):
        rc, so, se = out[r]
        assert rc == 0, (r, so, se)
        assert so == "OK " + EXPECT, (r, so)


def test_peers_time_out_when_rank0_never_comes():
    out = run_world(2, "tcp://127.0.0.1:%d" % free_port(), skip_rank0=True, timeout=1.5)
    rc, so, _ = out[1]
    assert rc == 3 and "could not fetch the id" in so


def test_rank0_times_out_when_a_peer_is_missing():
    import time
    from numpower_amd._lib import load
    lib = load()
    buf = (C.c_ubyte * 128)()
    t0 = time.time()
    rc = lib.np_comm_debug_exchange(0, 3, ("tcp://127.0.0.1:%d" % free_port()).encode(), buf, 1.5)
    assert rc != 0 and b"peers fetched the id" in lib.np_last_error()
    assert time.time() - t0 < 10


def test_a_stray_connection_is_not_counted_as_a_rank():
    """Something that is not a rank connects to rank 0's port first (and says nothing useful): it is dropped and the
    real peer is still served (rank 0 used to hand the id to any connector and count it)."""
    import threading
    import time
    port = free_port()
    endpoint = "tcp://127.0.0.1:%d" % port
    stop = threading.Event()

    def stray():
        # keeps knocking with garbage / silence until the exchange is over
        while not stop.is_set():
            try:
                with socket.create_connection(("127.0.0.1", port), timeout=0.2) as s:
                    s.sendall(b"GET / HTTP/1.0\r\n\r\n")
                    s.settimeout(0.2)
                    try:
                        s.recv(256)
                    except OSError:
                        pass
            except OSError:
                pass
            time.sleep(0.02)

    t = threading.Thread(target=stray, daemon=True)
    t.start()
    try:
        out = run_world(2, endpoint, rank0_delay=0.3, timeout=20.0)
    finally:
        stop.set()
        t.join(timeout=5)
    for r in range(2):
        rc, so, se = out[r]
        assert rc == 0, (r, so, se)
        assert so == "OK " + EXPECT


def test_a_stale_id_file_is_replaced_by_rank0(tmp_path):
    """A file left behind by a run that died after publishing must not reach the peers of the next run."""
    path = tmp_path / "id"
    path.write_bytes(bytes(128))          # a dead id of the right size
    out = run_world(2, str(path), rank0_delay=0.0)
    rc0, so0, se0 = out[0]
    assert rc0 == 0, (so0, se0)
    # the peer may have started polling before rank 0 replaced the file: what it must never do is hang; and a peer
    # that starts after rank 0 sees the fresh bytes
    out2 = run_world(2, str(path), rank0_delay=0.0, skip_rank0=True, timeout=5.0)
    rc1, so1, _ = out2[1]
    assert rc1 == 0 and so1 == "OK " + EXPECT
