"""N > 1 as separate processes.  (a) `python bench.py --gpus 2` with no launcher: the parent starts its two rank processes
itself (it never touches the GPU) and rank 0's JSON line comes back -- rehearsed here with the gloo backend, both ranks on
the one GPU of the test box, weak and strong scaling.  (b) two ranks over the real nccl (= RCCL) backend, one GPU each, stepping
SHMIP A3 -- and cfg5 on a hierarchy of box unions (ncclAllGather of the coarse cells level 1 reads) -- bit for bit against the
single-process run: skipped unless the box has two GPUs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_starts_its_own_ranks(scaling):
    env = dict(os.environ, SUHMO_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", GLOO_SOCKET_IFNAME="lo")   # (the box's hostname may not resolve)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "512", "--steps", "3", "--warmup", "1",
                        "--no-cpu", "--no-side", "--scaling", scaling], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["value"] > 0
    assert out["config"]["global_cells"] == ([512, 1024] if scaling == "weak" else [512, 512])
    assert out["config"]["halo_message_groups_per_vcycle_per_rank"] > 0
    assert out["roofline"]["traffic"] is None            # the PMC passes belong to the single-GPU workload
    assert out["residual_max_norm"]["after_timed_cycles"] < out["residual_max_norm"]["before_warmup"]


def test_bench_refuses_a_world_that_is_not_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "256", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-side"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE" in p.stderr


@pytest.mark.parametrize("tool,args", [("shmip_dist.py", ["--case", "A3", "--scale", "2", "--steps", "6", "--check"]),
                                       ("hier_dist.py", ["--base", "256", "--steps", "2", "--check"])])
def test_two_ranks_over_nccl_bitwise(tool, args):
    from suhmo_amd import capi
    if capi.lib().suhmo_device_count() < 2:
        pytest.skip("needs two GPUs (ncclSend / ncclRecv between distinct devices)")
    from conftest import free_port
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SUHMO_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", tool)] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    assert "BITWISE EQUAL" in logs[0]


@pytest.mark.parametrize("tool,args,world,agg", [("shmip_dist.py", ["--case", "A3", "--scale", "2", "--steps", "6", "--check", "--ipc"], 2, 0),
                                                 ("shmip_dist.py", ["--case", "B3", "--scale", "1", "--steps", "4", "--check", "--ipc"], 4, 3000),
                                                 ("hier_dist.py", ["--base", "256", "--steps", "2", "--check", "--ipc"], 2, 0),
                                                 ("shmip_dist.py", ["--case", "A3", "--scale", "2", "--steps", "4", "--check", "--ipc-probe"], 2, 0),
                                                 ("shmip_dist.py", ["--case", "A3", "--scale", "2", "--steps", "4", "--check", "--ipc-probe-fails"], 2, 0),
                                                 ("amr_shmip_dist.py", ["--case", "B5", "--steps", "6", "--check"], 2, 0),
                                                 ("shmip_dist.py", ["--case", "B3", "--scale", "1", "--steps", "6", "--check"], 4, 0),
                                                 ("hier_dist.py", ["--base", "256", "--steps", "2", "--check"], 2, 0),
                                                 ("shmip_dist.py", ["--case", "B3", "--scale", "1", "--steps", "6", "--check"], 4, 3000),
                                                 ("amr_shmip_dist.py", ["--case", "B5", "--steps", "6", "--check"], 2, 100000),
                                                 ("hier_dist.py", ["--base", "256", "--steps", "2", "--check"], 2, 10000),
                                                 ("hier_dist.py", ["--base", "256", "--steps", "2", "--check", "--partition-min-cells", "1"], 2, 0),
                                                 ("hier_dist.py", ["--base", "192", "--levels", "3", "--steps", "1", "--check", "--partition-min-cells", "1"], 3, 10000),
                                                 ("hier_dist.py", ["--base", "512", "--levels", "3", "--steps", "1", "--check", "--partition-min-cells", "1", "--dense"], 2, 0)])
def test_rank_strips_as_processes(tool, args, world, agg):
    """cfg4: SHMIP B5 (100 moulins, diffusion, implicit gap-height solve) on a 3-level AMR hierarchy cut into the strips of 2
    processes, B3 single-level on 4 processes, and cfg5 (base 256^2 + 3 levels of ~65 boxes each, 63 moulins) with level 0 cut into
    the strips of 2 processes and the boxes on both, or (--partition-min-cells 1) dealt to their owners on 2 and 3 processes; --dense: level 1 =
    64 boxes of 64^2 tiling the middle of a 512^2 base, so that neighbouring boxes, and fine boxes and the coarse boxes under them, have
    different owners (ghost cells, coarse-fine stencils, windows, flux registers and averages all cross ranks) (a job
    with thousands of small collectives stays at 3 ranks: with this process, which holds a GPU context of its own by then, a fifth
    process on the card makes every synchronisation of every rank wait for its turn -- measured: 7 s alone, > 300 s inside the suite)
    (gloo, all ranks on the one GPU of the test box): every level's head, gap height and melt rate equal the single-process
    run bit for bit (the tool's --check)"""
    from conftest import free_port
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SUHMO_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", SUHMO_AGG_MIN_CELLS=str(agg),    # agg > 0: coarse depths agglomerated (all-gather over gloo)
                   SUHMO_DUMP_AFTER="240", GLOO_SOCKET_IFNAME="lo")                                                            # a rank that hangs says where before it is killed
        if "--ipc" in args:            # the halo rows peer-direct between the PROCESSES (hipIpcGetMemHandle / hipIpcOpenMemHandle of each other's arenas on
            env["SUHMO_TRANSPORT"] = "ipc"     # the one GPU of the test box), reductions and all-gathers over gloo
        # --ipc-probe: what multigpu.attach does by default on the nccl backend: map, send three checked messages, keep the peer-direct path only if
        # every rank saw them intact; --ipc-probe-fails: rank 1 reports a failure, every rank goes back to the transport it had (and the run is still bitwise)
        if "--ipc-probe" in args or "--ipc-probe-fails" in args:
            env["SUHMO_TRANSPORT"] = "ipc-probe"
            if "--ipc-probe-fails" in args:
                env["SUHMO_IPC_PROBE_FAIL_RANK"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", tool)] + [a for a in args if not a.startswith("--ipc")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    import time
    deadline = time.time() + 300
    while time.time() < deadline and any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):      # a rank died: the others would wait for it until they time out
            time.sleep(2.0)
            break
        time.sleep(0.2)
    if any(p.poll() is None for p in procs):
        for q in procs:
            q.kill()
        tails = [q.communicate()[0].decode()[-3000:] for q in procs]
        pytest.fail(("a rank failed" if any(q.returncode not in (None, 0, -9) for q in procs) else "timed out") + "; the ranks' output:\n" + "\n-----\n".join(tails))
    logs = [p.communicate()[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    assert "BITWISE EQUAL" in logs[0]
    if "--ipc-probe" in args:
        assert "three probe messages arrived intact on every rank -> ipc" in logs[0], logs[0][-1500:]
    if "--ipc-probe-fails" in args:
        assert "peer-direct transport not available" in logs[0] and "-> the transport it had" in logs[0], logs[0][-1500:]
