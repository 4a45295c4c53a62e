"""Which host memory feeds the link fastest? Pinned H2D rate with the allocating thread bound to each NUMA node, alone and with reader threads busy."""
import glob, os, time, threading
import numpy as np
import torch

def cpus_of(node):
    s = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
    out = []
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out

print("affinity now:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes:", nodes)
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
    try:
        print(p, open(p).read().strip(), open(os.path.dirname(p) + "/vendor").read().strip())
    except OSError as e:
        print(p, e)
allowed = os.sched_getaffinity(0)
dev = torch.device("cuda:0")
d = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
for node in nodes:
    cp = [c for c in cpus_of(node) if c in allowed]
    if not cp:
        print("node", node, "no allowed cpus"); continue
    os.sched_setaffinity(0, cp)
    h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    h.fill_(1)
    torch.cuda.synchronize()
    best = 0
    for _ in range(5):
        t = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
        best = max(best, (1 << 30) / dt / 1e9)
    # with 8 threads copying host memory on the same node
    stop = False
    src = np.ones(1 << 28, dtype=np.uint8); dsts = [np.empty(1 << 28, dtype=np.uint8) for _ in range(8)]
    def work(i):
        while not stop:
            np.copyto(dsts[i], src)
    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    time.sleep(0.2)
    busy = 0
    for _ in range(5):
        t = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
        busy = max(busy, (1 << 30) / dt / 1e9)
    stop = True
    [t.join() for t in th]
    print(f"node {node}: {len(cp)} cpus, pinned H2D {best:.1f} GB/s alone, {busy:.1f} GB/s with 8 memcpy threads on the node")
    del h
    os.sched_setaffinity(0, allowed)
