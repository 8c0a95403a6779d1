"""Diagnostic for DESIGN.md 6a': which resource of a long-lived parent makes NEW GPU processes die in hsa_init.

Runs `diffusion_impute` calls like tests/test_gpu_folds.py and after each one (a) snapshots what the parent holds
(live hipGraph execs, mappings by kind, descriptors by kind, KFD per-process sysfs, /dev/shm, host-memory counters) and
(b) starts scripts/diag/hsa_probe (hsa_init + a back-trace on a fault) as a child.  At the first failing probe it tries
the releases one by one (gc, empty_cache, unloading nothing) with a probe after each, then child variants with
HSAKMT_DEBUG_LEVEL / HSA_ENABLE_INTERRUPT=0 / HSA_ENABLE_SDMA=0.

    python scripts/diag/impute_leak.py <dtypes> <spots> <steps> <folds> [lanes]
"""
import collections
import gc
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as O  # noqa: E402
from stdiff_amd import ops  # noqa: E402

PROBE = os.path.join(ROOT, "scripts", "diag", "hsa_probe")
LIVE = {"graphs": 0, "graphs_made": 0, "launches": 0}
_end, _del, _launch = ops.Graph.end, ops.Graph.__del__, ops.Graph.launch


def end(self):
    _end(self)
    LIVE["graphs"] += 1
    LIVE["graphs_made"] += 1


def dele(self):
    if self.exec is not None:
        LIVE["graphs"] -= 1
    _del(self)


def launch(self):
    LIVE["launches"] += 1
    _launch(self)


ops.Graph.end, ops.Graph.__del__, ops.Graph.launch = end, dele, launch


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError as e:
        return f"<{e.__class__.__name__}>"


def snapshot():
    s = dict(LIVE)
    maps = read("/proc/self/maps").splitlines()
    kinds = collections.Counter()
    size = collections.Counter()
    for ln in maps:
        parts = ln.split()
        name = parts[5] if len(parts) > 5 else "anon"
        key = ("kfd" if "kfd" in name else "dri" if "dri" in name else "shm" if "/dev/shm" in name or "memfd" in name
               else "lib" if ".so" in name else "anon" if name == "anon" else "other")
        a, b = parts[0].split("-")
        kinds[key] += 1
        size[key] += int(b, 16) - int(a, 16)
    s["maps"] = len(maps)
    s["maps_kind"] = dict(kinds)
    s["maps_GiB"] = {k: round(v / 2 ** 30, 2) for k, v in size.items()}
    fds = collections.Counter()
    for fd in os.listdir("/proc/self/fd"):
        try:
            t = os.readlink(f"/proc/self/fd/{fd}")
        except OSError:
            continue
        fds["kfd" if "kfd" in t else "dri" if "dri" in t else "eventfd" if "eventfd" in t else "shm" if "shm" in t
            else "sock/pipe" if t.startswith(("socket", "pipe")) else "other"] += 1
    s["fds"] = dict(fds)
    st = read("/proc/self/status")
    s["status"] = {k: v.strip() for k, v in (ln.split(":", 1) for ln in st.splitlines() if ":" in ln)
                   if k in ("VmRSS", "VmLck", "VmPin", "Threads", "VmSize")}
    mi = read("/proc/meminfo")
    s["meminfo"] = {k: v.strip() for k, v in (ln.split(":", 1) for ln in mi.splitlines() if ":" in ln)
                    if k in ("MemAvailable", "Shmem", "Unevictable", "Mlocked", "Mapped")}
    try:
        v = os.statvfs("/dev/shm")
        s["dev_shm_used_MiB"] = round((v.f_blocks - v.f_bfree) * v.f_frsize / 2 ** 20, 1)
        s["dev_shm_total_MiB"] = round(v.f_blocks * v.f_frsize / 2 ** 20, 1)
        s["dev_shm_files"] = len(os.listdir("/dev/shm"))
    except OSError as e:
        s["dev_shm"] = repr(e)
    kp = f"/sys/class/kfd/kfd/proc/{os.getpid()}"
    if os.path.isdir(kp):
        ent = {}
        for n in sorted(os.listdir(kp)):
            p = os.path.join(kp, n)
            ent[n] = (len(os.listdir(p)) if os.path.isdir(p) else read(p).strip()[:40])
        s["kfd_proc"] = ent
    else:
        s["kfd_proc"] = "absent"
    s["kfd_procs_total"] = len(os.listdir("/sys/class/kfd/kfd/proc")) if os.path.isdir("/sys/class/kfd/kfd/proc") else -1
    s["sysv"] = {k: max(0, len(read(f"/proc/sysvipc/{k}").splitlines()) - 1) for k in ("shm", "sem")}
    s["torch_MiB"] = (torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, torch.cuda.mem_get_info()[0] >> 20)
    return s


def probe(tag, env=None, show=False):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([PROBE], capture_output=True, text=True, env=e)
    ok = r.returncode == 0
    print(f"PROBE {tag}: rc {r.returncode} {r.stdout.strip()}", flush=True)
    if not ok or show:
        print("   stderr tail:\n      " + "\n      ".join(r.stderr.strip().splitlines()[-60:]), flush=True)
    return ok


def main():
    dts, N, T, nfold = sys.argv[1].split(","), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    if len(sys.argv) > 5:
        os.environ["STDIFF_LANES"] = sys.argv[5]
    from stdiff_amd import driver
    from stdiff_amd.driver import diffusion_impute
    if os.environ.get("DIAG_EAGER") == "1":               # no hipGraph: every reverse step enqueued kernel by kernel
        import functools
        driver.sample_stDiff = functools.partial(driver.sample_stDiff, use_graph=False)
    G, H, depth, heads, B = 500, 1024, 6, 16, 512
    seq, st = O.synth_expression(1024, G, 21), O.synth_expression(N, G, 22)
    print("before any GPU call:", snapshot(), flush=True)
    probe("before any GPU call")
    torch.ones(4, device="cuda:0")
    print("after GPU init:", snapshot(), flush=True)
    probe("after GPU init")
    failed_at = None
    with tempfile.TemporaryDirectory() as ck:
        for k in range(5):
            torch.save(O.make_params(G, H, depth, 60 + k), f"{ck}/stDiff{k + 1}.pt")
        n = 0
        for dt in dts:
            for k in range(nfold):
                r = diffusion_impute(seq, st, depth=depth, num_epoch=1, diffusion_step=T, batch_size=B, hidden_size=H,
                                     head=heads, device=torch.device("cuda:0"), save_dir=ck, sample_seed=1,
                                     compute_dtype=dt, folds=[k])
                n += 1
                print(f"call {n} ({dt} fold {k}) sum {float(np.abs(r).sum()):.1f}:", snapshot(), flush=True)
                if not probe(f"after call {n}"):
                    failed_at = n
                    break
            if failed_at:
                break
    if not failed_at:
        print("RESULT: no probe failed", flush=True)
        return
    print(f"RESULT: first failing probe after call {failed_at}", flush=True)
    probe("again, same state")
    if os.environ.get("DIAG_VERBOSE") == "1":
        e = dict(os.environ, HSAKMT_DEBUG_LEVEL="7")
        r = subprocess.run([PROBE], capture_output=True, text=True, env=e)
        print("FULL KMT DEBUG of a failing child:\n" + r.stderr, flush=True)
        for cmd in (["dmesg"], ["journalctl", "-k", "-n", "50", "--no-pager"]):
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
                print(" ".join(cmd), "rc", r.returncode, "\n", (r.stdout + r.stderr)[-3000:], flush=True)
            except Exception as ex:    # noqa: BLE001
                print(" ".join(cmd), "failed:", repr(ex), flush=True)
    if os.environ.get("DIAG_SHORT") == "1":
        return
    probe("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"})
    probe("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"})
    probe("HSA_ENABLE_IPC_MODE_LEGACY unset", {"HSA_ENABLE_IPC_MODE_LEGACY": ""})
    torch.cuda.synchronize()
    probe("after synchronize")
    gc.collect()
    print("after gc:", snapshot(), flush=True)
    probe("after gc.collect")
    torch.cuda.empty_cache()
    print("after empty_cache:", snapshot(), flush=True)
    probe("after empty_cache")
    r = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True)
    print("rocminfo rc", r.returncode, r.stderr.strip()[-300:], flush=True)


if __name__ == "__main__":
    main()
