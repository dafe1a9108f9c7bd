"""Host-side hygiene of a one-process-per-GPU run: thread caps and a core set per rank.

The reference starts its ranks with detectron2's ``launch()`` (train_net.py:132-139) and leaves the host to the defaults; on an
8-GPU MI355X node that is 8 processes x a 256-thread intra-op pool, all free to migrate over both sockets, on a step whose
target-detector flavour is host-sensitive.  This module gives every rank

* a disjoint set of cores, NUMA-local where the topology is readable (``/sys/devices/system/node``): the node's physical cores
  are divided among the ranks that are placed on it, SMT siblings travel with their core;
* ``OMP_NUM_THREADS`` / ``MKL_NUM_THREADS`` / ``torch.set_num_threads`` capped to that set (at most ``MAX_THREADS``).

Nothing here touches the GPU, re-executes the process or imports torch at module import: call ``apply_rank_affinity`` first thing in
a rank (bench.py, train_net.py), before the HIP runtime starts.  ``COIN_RANK_CPUS`` overrides the computed sets
(``"0-15;16-31;..."``: one cpulist per local rank); ``COIN_RANK_AFFINITY=0`` switches the pinning off (thread caps stay).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Set

MAX_THREADS = 8   # intra-op host threads per rank: the step's host work is one Python thread + short CPU tensor ops


def parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def format_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(set(int(c) for c in cpus))
    parts, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(parts)


def read_topology(sysfs: str = "/sys/devices/system") -> Dict:
    """-> {"nodes": {node id: [cpus]}, "siblings": {cpu: (cpus of its core)}} from sysfs; empty dicts where it is not readable."""
    nodes: Dict[int, List[int]] = {}
    siblings: Dict[int, tuple] = {}
    try:
        for name in sorted(os.listdir(os.path.join(sysfs, "node"))):
            if name.startswith("node") and name[4:].isdigit():
                with open(os.path.join(sysfs, "node", name, "cpulist")) as f:
                    cpus = parse_cpulist(f.read())
                if cpus:
                    nodes[int(name[4:])] = cpus
    except OSError:
        pass
    try:
        for name in os.listdir(os.path.join(sysfs, "cpu")):
            if name.startswith("cpu") and name[3:].isdigit():
                try:
                    with open(os.path.join(sysfs, "cpu", name, "topology", "thread_siblings_list")) as f:
                        siblings[int(name[3:])] = tuple(sorted(parse_cpulist(f.read())))
                except OSError:
                    pass
    except OSError:
        pass
    return {"nodes": nodes, "siblings": siblings}


def rank_cpu_sets(local_world: int, allowed: Optional[Set[int]] = None, topology: Optional[Dict] = None) -> List[List[int]]:
    """Disjoint core sets for `local_world` ranks of one machine.  Ranks are placed on the NUMA nodes in blocks (ranks
    [0, ceil(W / nodes)) on the first node, ...: GPUs 0-3 hang off socket 0 and 4-7 off socket 1 on the 8-GPU MI300X / MI355X platforms;
    `COIN_RANK_CPUS` overrides where a machine differs); a node's cores -- whole cores, SMT siblings together -- are dealt out in
    contiguous runs to its ranks.  Every rank gets at least one cpu as long as there are at least `local_world` cores."""
    if allowed is None:
        allowed = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    topo = topology if topology is not None else read_topology()
    sib = topo.get("siblings") or {}
    node_cpus = [sorted(set(c) & allowed) for _, c in sorted((topo.get("nodes") or {}).items())]
    node_cpus = [c for c in node_cpus if c]
    if not node_cpus or set().union(*map(set, node_cpus)) != allowed:
        node_cpus = [sorted(allowed)]   # topology unreadable or inconsistent with the allowed set: one pool

    def cores_of(cpus: List[int]) -> List[List[int]]:
        seen, cores = set(), []
        cs = set(cpus)
        for c in cpus:
            if c in seen:
                continue
            grp = [s for s in sib.get(c, (c,)) if s in cs] or [c]
            seen.update(grp)
            cores.append(sorted(grp))
        return cores

    n_nodes = min(len(node_cpus), local_world)
    per_node = -(-local_world // n_nodes)
    out: List[List[int]] = []
    for r in range(local_world):
        node = min(r // per_node, n_nodes - 1)
        first = node * per_node
        ranks_here = min(per_node, local_world - first)
        cores = cores_of(node_cpus[node])
        k = r - first
        lo, hi = k * len(cores) // ranks_here, (k + 1) * len(cores) // ranks_here
        mine = cores[lo:hi] or cores[min(lo, len(cores) - 1):min(lo, len(cores) - 1) + 1]
        out.append(sorted(c for core in mine for c in core))
    return out


def rank_threads(cpus: Sequence[int], topology: Optional[Dict] = None) -> int:
    """Intra-op threads for a rank that owns `cpus`: one per physical core, at most MAX_THREADS."""
    sib = (topology if topology is not None else read_topology()).get("siblings") or {}
    cores = {sib.get(c, (c,))[0] for c in cpus}
    return max(1, min(MAX_THREADS, len(cores)))


def rank_env(local_rank: int, local_world: int, base: Optional[Dict[str, str]] = None, allowed: Optional[Set[int]] = None,
             topology: Optional[Dict] = None) -> Dict[str, str]:
    """Environment additions for local rank `local_rank` of `local_world`: thread caps + the core set it will pin itself to
    (``COIN_RANK_CPUSET``; `apply_rank_affinity` in the rank reads it).  A value already present in `base` wins."""
    base = base if base is not None else dict(os.environ)
    override = base.get("COIN_RANK_CPUS")
    if override:
        lists = [parse_cpulist(x) for x in override.split(";")]
        cpus = lists[local_rank % len(lists)]
    else:
        cpus = rank_cpu_sets(local_world, allowed, topology)[local_rank]
    n = rank_threads(cpus, topology)
    env = {"COIN_RANK_CPUSET": format_cpulist(cpus)}
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        env[k] = base.get(k) or str(n)
    return env


def apply_rank_affinity(local_rank: Optional[int] = None, local_world: Optional[int] = None, cpu_compute: bool = False) -> Dict:
    """In a rank, before the HIP runtime starts: pin this process to its core set and cap the host thread pools.
    A single-rank run (`local_world` 1) is left alone apart from an OMP cap of MAX_THREADS when none is set (not even that with
    `cpu_compute`: a run whose model lives on the CPU).
    -> {"cpus": cpulist or None, "threads": n}: what was applied (bench.py echoes it as `config.rank_affinity`)."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    if local_world is None:
        # ranks on THIS node: the launcher's LOCAL_WORLD_SIZE; without it (srun / mpirun style launches) the GPUs visible here -- never the global
        # WORLD_SIZE, which on two nodes of eight would hand every rank a sixteenth of the cores, all of them on the first socket (round-5 ADVICE)
        lw = os.environ.get("LOCAL_WORLD_SIZE")
        if lw is not None:
            local_world = int(lw)
        elif int(os.environ.get("WORLD_SIZE", "1")) <= 1:
            local_world = 1
        else:
            try:
                import torch   # device_count() does not initialise the HIP runtime on this image

                local_world = max(1, min(int(os.environ["WORLD_SIZE"]), torch.cuda.device_count()))
            except Exception:
                local_world = int(os.environ["WORLD_SIZE"])
    applied: Dict = {"cpus": None, "threads": None}
    if local_world <= 1:
        if cpu_compute:   # MODEL.DEVICE cpu (the parity / plumbing mode of train_net.py): the host pools ARE the compute, leave them alone
            return applied
        os.environ.setdefault("OMP_NUM_THREADS", str(MAX_THREADS))
        applied["threads"] = int(os.environ["OMP_NUM_THREADS"])
        return applied
    preset = os.environ.get("COIN_RANK_CPUSET")
    env = rank_env(local_rank, local_world) if not preset else {}
    cpus = parse_cpulist(preset or env["COIN_RANK_CPUSET"])
    # torch.distributed.run exports OMP_NUM_THREADS=1 for every rank it starts; a value the user set is respected the same way
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(k, env.get(k, str(rank_threads(cpus))))
    applied["threads"] = int(os.environ["OMP_NUM_THREADS"])
    if os.environ.get("COIN_RANK_AFFINITY", "1") != "0" and hasattr(os, "sched_setaffinity") and cpus:
        try:
            os.sched_setaffinity(0, set(cpus) & set(os.sched_getaffinity(0)) or set(cpus))
            applied["cpus"] = format_cpulist(os.sched_getaffinity(0))
        except OSError:
            pass
    return applied


def cap_torch_threads(applied: Dict) -> None:
    """After `import torch`: the intra-op pool follows the cap (the OMP variable alone does not bind a pool torch already sized)."""
    import torch

    if applied.get("threads"):
        torch.set_num_threads(int(applied["threads"]))
