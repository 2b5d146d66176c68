"""Pin a rank's host threads to the NUMA node its GPU hangs off.

One process drives one MI355X.  At 0.1 ms per step (configs[1]-sized shards) the host side
of a step — a handful of kernel launches — is latency sensitive: a rank whose launch thread runs on the
other socket pays a cross-socket hop per doorbell write.  The GPU's node comes from sysfs through the
PCI bus id the library reports (`f110_device_pci_bus_id` = hipDeviceGetPCIBusId); the binding itself is
`os.sched_setaffinity` on the whole process (stdlib; no libnuma, no torch).
"""
import ctypes as C
import os

from . import _ffi


def device_pci_bus_id(device_id):
    buf = C.create_string_buffer(64)
    _ffi.check(_ffi.lib().f110_device_pci_bus_id(int(device_id), buf, 64), None)
    return buf.value.decode().lower()


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pci_numa_node(bus_id, sysfs="/sys"):
    """NUMA node of a PCI device, or -1 when the platform does not say (single-node boxes, VMs)"""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", bus_id, "numa_node")) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def node_cpus(node, sysfs="/sys"):
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
            return parse_cpulist(f.read())
    except (OSError, ValueError):
        return []


def bind_to_node_of(bus_id, sysfs="/sys", setaffinity=None, allowed=None):
    """-> {"pci": ..., "numa_node": n, "cpus_bound": k or None, "note": ...}; never raises"""
    rec = {"pci": bus_id, "numa_node": pci_numa_node(bus_id, sysfs), "cpus_bound": None}
    if rec["numa_node"] < 0:
        rec["note"] = "sysfs reports no NUMA node for this device: affinity left as it was"
        return rec
    cpus = node_cpus(rec["numa_node"], sysfs)
    if allowed is None:
        allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(cpus)
    cpus = sorted(set(cpus) & set(allowed))   # stay inside the cpuset the launcher / container granted
    if not cpus:
        rec["note"] = "no allowed CPU on node %d: affinity left as it was" % rec["numa_node"]
        return rec
    try:
        (setaffinity or (lambda c: os.sched_setaffinity(0, c)))(cpus)
        rec["cpus_bound"] = len(cpus)
    except OSError as ex:
        rec["note"] = "sched_setaffinity failed: %s" % ex
    return rec


def bind_to_device(device_id):
    """bind the calling process to the CPUs of HIP device `device_id`'s NUMA node"""
    try:
        return bind_to_node_of(device_pci_bus_id(device_id))
    except Exception as ex:  # noqa: BLE001 — a missing sysfs entry must never stop a run
        return {"pci": None, "numa_node": -1, "cpus_bound": None, "note": "%s" % ex}
