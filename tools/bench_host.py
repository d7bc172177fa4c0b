"""Host-side support of bench.py, kept out of the benchmark file itself: the CPU baseline (the reference's compiled ops + the torch fp32
restatement on the box's host cores, in fresh worker processes) and the rank launch / core binding of the N-GPU form.  bench.py imports
from here; `python bench.py --cpu-worker ...` is still the worker command line (it dispatches to `cpu_worker`)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
VOXEL, RADIUS, NUM_STAGES = 0.3, 1.275, 4
LIMITS = [64, 65, 74, 80]          # reference training/eval default (dataset_loop_detection.py:25,80)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the same path on the host cores of this box.  SURVEY §8d: pre-processing only and end to end,
# at 1 core and at all cores with ONE SCAN PER PROCESS (pre-processing: one single-thread process per core; end to end: one
# 8-thread process per 8 cores) (the reference's own parallelism: DataLoader worker processes,
# config_ld.py:42).  Native ops = the reference's C++ compiled from source (oracle/_ref) when present, else the C++ restatement;
# encoder + NetVLAD = the torch fp32 restatement.  Each worker is a fresh interpreter (no fork of a process that owns a GPU).
# ---------------------------------------------------------------------------------------------------------------------
def cpu_worker(mode, scan_path, threads, start_at):
    """One scan, `mode` in {pre, e2e}, on `threads` torch threads, starting at wall-clock `start_at`.  Prints seconds from
    start_at to completion (a worker that is not ready in time is charged for its lateness)."""
    threads = int(threads)
    if mode == "e2e":
        import torch
        torch.set_num_threads(threads)
    from oracle import ops as oracle_ops
    impl = "ref" if oracle_ops.have_ref() else "oracle"
    raw = np.load(scan_path)
    sd = None
    if mode == "e2e":
        from oracle import torch_ref
        from lcrnet_amd.model_family import create_model
        from lcrnet_amd.weights import seeded_state_dict
        sd = seeded_state_dict(create_model().state_dict(), 7351)
    oracle_ops.grid_subsample(raw[:1000], np.array([1000]), VOXEL, impl=impl)     # library loaded before the clock starts
    start_at = float(start_at)
    while time.time() < start_at:
        time.sleep(0.001)
    p, l = oracle_ops.grid_subsample(raw, np.array([len(raw)]), VOXEL, impl=impl)
    st = oracle_ops.precompute_data_stack_mode(p, l, NUM_STAGES, VOXEL, RADIUS, LIMITS, impl=impl)
    if mode == "e2e":
        with torch.no_grad():
            dd = {k: [torch.from_numpy(np.ascontiguousarray(x)) for x in v] for k, v in st.items()}
            feats = torch_ref.kp_encoder(sd, torch.ones(len(p), 1), dd)
            torch_ref.global_descriptor(sd, feats[-1])
    print(json.dumps({"seconds": time.time() - start_at, "impl": impl}))


def _run_cpu_workers(mode, scan_paths, procs, threads, lead):
    start_at = time.time() + lead
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    ps = [subprocess.Popen([sys.executable, BENCH, "--cpu-worker", mode, scan_paths[i % len(scan_paths)], str(threads),
                            repr(start_at)], stdout=subprocess.PIPE, env=env, cwd=ROOT) for i in range(procs)]
    outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in ps]
    wall = max(o["seconds"] for o in outs)
    return procs / wall, wall, outs[0]["impl"]


def cpu_baseline(scans):
    import tempfile
    cores = os.cpu_count() or 1
    try:
        import psutil
        avail_gb = psutil.virtual_memory().available / 2**30
    except Exception:
        avail_gb = 64.0
    tmp = tempfile.mkdtemp(prefix="lcr_cpu_")
    paths = []
    for i, s in enumerate(scans):
        paths.append(os.path.join(tmp, "scan%d.npy" % i))
        np.save(paths[-1], s)
    lead1 = 6.0                                                       # interpreter + torch import + weights before the clock starts
    leadN = 8.0 + 0.15 * cores
    pre1, _, impl = _run_cpu_workers("pre", paths, 1, 1, 2.0)        # numpy + ctypes only: short lead
    preN, pre_wall, _ = _run_cpu_workers("pre", paths, cores, 1, 3.0 + 0.05 * cores)
    e1, e1_wall, _ = _run_cpu_workers("e2e", paths, 1, 1, lead1)
    # end to end on all cores: processes of 8 torch threads each, one scan per process — the reference's own shape (torch intra-op
    # threads for the model next to its DataLoader workers).  Measured on the 256-core GPU box: 256 single-thread processes reach
    # only 2.7 scans/s (5.6x one core: the fp32 encoder is memory-bound and 256 private copies of its ~1.5 GB of intermediates
    # thrash the caches), so that split is not used.
    thr = min(8, cores)
    procs = int(max(1, min(cores // thr, avail_gb // 3)))
    eN, eN_wall, _ = _run_cpu_workers("e2e", paths, procs, thr, leadN)
    for p in paths:
        os.remove(p)
    os.rmdir(tmp)
    kind = "reference" if impl == "ref" else "port"
    return {"value": round(eN, 4), "unit": "scans/s", "cores": cores, "kind": kind,
            "sample": "end to end (voxelise + collate + encoder + NetVLAD), all %d cores: %d processes x %d torch thread(s), one scan each, "
                      "%.1f s wall.  native ops: %s; encoder+NetVLAD: torch fp32 restatement.  Baseline only: the torch fp32 encoder is memory-bound "
                      "(all cores reach only ~4.5x one core), this is the reference's own CPU shape, not a tuned CPU implementation" %
                      (cores, procs, thr, eN_wall, "reference C++ compiled from source (oracle/_ref)" if impl == "ref" else "oracle C++ restatement"),
            "end_to_end": {"one_core_scans_per_s": round(e1, 4), "all_cores_scans_per_s": round(eN, 4), "processes": procs, "threads_per_process": thr,
                           "one_core_s_per_scan": round(e1_wall, 3)},
            "preprocessing_only": {"one_core_scans_per_s": round(pre1, 4), "all_cores_scans_per_s": round(preN, 4), "processes": cores,
                                   "all_cores_wall_s": round(pre_wall, 3),
                                   "what": "0.3 m voxelisation + 3 subsamples + 10 radius searches of one 120k-pt scan, one scan per process"}}


# ---------------------------------------------------------------------------------------------------------------------
def _cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(index):
    """NUMA node of GPU `index` from sysfs (PCI address from the device properties), or None."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(index)
        addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % addr).read())
        return node if node >= 0 else None
    except Exception:
        return None


def rank_cpus(local, n_local, single_device=False):
    """The host cores of local rank `local` of `n_local`: the cores of its GPU's NUMA node, shared out among the ranks whose GPUs sit
    on that node; without NUMA information an equal contiguous share of the cores this process may use.  Every rank runs ~6 busy
    host threads (2 pre-processing issuers, 2 encoder issuers, the consumer, RCCL's proxy): they should neither migrate across
    sockets nor pile onto one rank's cores."""
    allowed = sorted(os.sched_getaffinity(0))
    nodes = [None if single_device else gpu_numa_node(r) for r in range(n_local)]
    me = nodes[local]
    if me is not None:
        try:
            node_cpus = [c for c in _cpulist(open("/sys/devices/system/node/node%d/cpulist" % me).read()) if c in set(allowed)]
            peers = [r for r in range(n_local) if nodes[r] == me]
            k = peers.index(local)
            share = len(node_cpus) // len(peers)
            if share >= 1:
                return node_cpus[k * share:(k + 1) * share], "numa node %d, share %d/%d" % (me, k + 1, len(peers))
        except Exception:
            pass
    share = max(1, len(allowed) // n_local)
    return allowed[(local * share) % len(allowed):][:share], "even split of %d cores" % len(allowed)


def bind_rank(local, n_local, single_device=False):
    """Pin this rank process (and the threads it starts later) to its cores and cap the math libraries' thread pools."""
    import torch
    cpus, how = rank_cpus(local, n_local, single_device)
    try:
        os.sched_setaffinity(0, cpus)
    except Exception as e:                                             # containers may forbid it: not fatal
        how = "not bound (%s)" % e
    torch.set_num_threads(max(1, min(8, len(cpus))))                    # torch intra-op pool (host-side tensor ops are tiny here)
    return len(cpus), how


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (one per GPU) with the torchrun environment
    contract and wait for them; rank 0's JSON line goes to this process's stdout."""
    import socket
    import torch
    n = args.gpus
    if not os.environ.get("LCR_BENCH_SINGLE_DEVICE"):
        have = torch.cuda.device_count()
        if have < n:
            sys.exit("bench.py: --gpus %d but %d GPU(s) visible (LCR_BENCH_SINGLE_DEVICE=1 runs every rank on GPU 0 as a dry run)" % (n, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("OMP_NUM_THREADS", "8")                          # before the rank imports torch / numpy
        env.setdefault("MKL_NUM_THREADS", "8")
        # rank 0's JSON line is the only thing on stdout; the other ranks keep stderr
        procs.append(subprocess.Popen([sys.executable, BENCH] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            rc = p.wait(timeout=float(os.environ.get("LCR_BENCH_RANK_TIMEOUT", "3600"))) or rc
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:                                                 # a rank that died must not leave its peers in a barrier
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


