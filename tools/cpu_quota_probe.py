"""How many CPUs does this box really give a process?  sched_getaffinity counts the logical CPUs the process may run on; a container's CPU
quota (cgroup cpu.max / cfs_quota) caps the CPU TIME per period below that.  Prints both and measures it: N busy threads (pure integer work
in C through the oracle library would do; here: Python processes spinning) for a fixed wall time, total CPU seconds obtained / wall seconds."""
import multiprocessing as mp, os, time, resource, sys

def read(path):
    try:
        return open(path).read().strip()
    except OSError as e:
        return "n/a (%s)" % e.__class__.__name__

def spin(seconds, q):
    t0 = time.perf_counter(); c0 = time.process_time(); x = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20000): x += 1
    q.put(time.process_time() - c0)

if __name__ == "__main__":
    print("logical CPUs (sched_getaffinity): %d; os.cpu_count: %s" % (len(os.sched_getaffinity(0)), os.cpu_count()))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
        print("%s: %s" % (p, read(p).replace("\n", " | ")))
    for n in (1, 8, 16, 32, 64, 128):
        if n > len(os.sched_getaffinity(0)): break
        q = mp.Queue(); ps = [mp.Process(target=spin, args=(2.0, q)) for _ in range(n)]
        t0 = time.perf_counter()
        for p in ps: p.start()
        got = [q.get() for _ in ps]
        for p in ps: p.join()
        wall = time.perf_counter() - t0
        print("%4d busy processes for 2 s: %.1f CPU-seconds in %.2f s wall = %.1f CPUs' worth" % (n, sum(got), wall, sum(got) / 2.0), flush=True)
    print("cpu.stat after: %s" % read("/sys/fs/cgroup/cpu.stat").replace("\n", " | "))
