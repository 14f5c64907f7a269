"""Builds libminrms_hip.so (the C-ABI engine) in-tree with hipcc for gfx950.

The DP kernels (csrc/dp_kernel.h) are hand-written gfx950 assembly inside
compiler-scheduled code: they pin scalar registers s[83:99] by constraint,
rewrite `exec` by hand and rely on a whole row of costs staying in vector
registers, and the cooperative kernel counts its own vector-memory operations
(`s_waitcnt vmcnt(<operations of one row>)`).  A spill to scratch memory breaks
all of that silently -- scratch traffic is vector-memory traffic -- so the build
reads the compiler's own per-kernel resource report
(-Rpass-analysis=kernel-resource-usage) and fails if any DP kernel uses scratch
memory or spills a vector register, or if a sweep kernel (k_dp_panel) or a
cooperative kernel of the hand-written row loop (k_dp_coop<.., true, true>)
spills a scalar register at all.  The cooperative kernels of the compiled loop
(six variants of it in one kernel where a workgroup holds whole layer blocks)
may park scalar values in the lanes of ONE vector register (v_writelane /
v_readlane: no memory, independent of exec), no more.
The per-kernel table is written to profiles/kernel_resources.csv.
"""
import os
import re
import subprocess
import subprocess as _sp  # noqa: F811

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libminrms_hip.so")
# one object per unit, compiled side by side; the DP kernels (the bulk of the compile time) have units of their own
SOURCES = ["engine_core.hip", "search.hip", "refine.hip", "alternates.hip", "orientations.hip", "noalloc.hip",
           "dp_sweep_kernels.hip", "dp_coop_kernels.hip"]
# (source, object name, extra flags); the longest compiles first
UNITS = [("dp_sweep_kernels.hip", "dp_sweep_kernels_2.o", ["-DMINRMS_SWEEP_PART=2"]),
         ("dp_sweep_kernels.hip", "dp_sweep_kernels_1.o", ["-DMINRMS_SWEEP_PART=1"]),
         ("dp_sweep_kernels.hip", "dp_sweep_kernels_0.o", ["-DMINRMS_SWEEP_PART=0"]),
         ("dp_coop_kernels.hip", "dp_coop_kernels.o", [])] + \
        [(s, s.replace(".hip", ".o"), []) for s in ("search.hip", "refine.hip", "alternates.hip", "engine_core.hip",
                                                    "orientations.hip", "noalloc.hip")]
HEADERS = ["engine_internal.h", "dp_kernel.h", "dp_task_asm.h", "fit_device.h", "layout.h", "nw_filter.h",
           "nw_filter_params.h", "alt_solutions.h", "jacobi_sym.h", "../../include/minrms_engine.h"]
RESOURCE_TABLE = os.path.join(ROOT, "profiles", "kernel_resources.csv")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # float results must carry the reference's roundings: never fuse a*b+c
    "-ffp-contract=off",
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    "-Wno-unused-command-line-argument", "-Wno-pass-failed",
    "-Rpass-analysis=kernel-resource-usage",
]

# gfx950: 102 addressable SGPRs + VCC; TotalSGPRs of the report includes VCC and
# the 6 reserved for FLAT_SCRATCH / XNACK_MASK
MAX_TOTAL_SGPRS = 108
# kernel name pattern (regular expression) -> scalar values it may park in vector-register lanes (first match counts): the
# sweep kernels and the cooperative kernels of the hand-written row loop (third template argument true) none
GUARDED = [(r"k_dp_panel", 0), (r"k_dp_coop<\d+, true, true>", 0), (r"k_dp_coop", 64)]


class BuildError(RuntimeError):
    pass


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            p = _sp.run([tool], input="\n".join(names) + "\n", stdout=_sp.PIPE, text=True)
            out = p.stdout.split("\n")
            if p.returncode == 0 and len(out) >= len(names):
                return out[:len(names)]
        except OSError:
            pass
    return names


def parse_resource_report(text):
    """[{name, sgprs, vgprs, agprs, scratch, occupancy, sgpr_spill, vgpr_spill, lds}] from hipcc's remarks."""
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
            "LDS Size [bytes/block]": "lds"}
    kernels, cur = [], None
    for line in text.split("\n"):
        m = re.search(r"remark:\s+(.*?): (\S+) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"mangled": v}
            kernels.append(cur)
        elif cur is not None and k in keys:
            cur[keys[k]] = int(v)
    for k, name in zip(kernels, _demangle([k["mangled"] for k in kernels])):
        k["name"] = re.sub(r"^void ", "", re.sub(r"\(.*$", "", name))   # template arguments stay, the parameter list goes
    return kernels


def check_resources(kernels):
    """Raises BuildError if a DP kernel spills, uses scratch or runs out of scalar registers."""
    bad = []
    seen = 0
    for k in kernels:
        allowed = [v for g, v in GUARDED if re.search(g, k["name"])]
        if not allowed:
            continue
        seen += 1
        if (k.get("scratch", 0) or k.get("sgpr_spill", 0) > allowed[0] or k.get("vgpr_spill", 0)
                or k.get("sgprs", 0) > MAX_TOTAL_SGPRS):
            bad.append("%s: scratch %d B/lane, SGPR spills %d, VGPR spills %d, TotalSGPRs %d" % (
                k["name"], k.get("scratch", 0), k.get("sgpr_spill", 0), k.get("vgpr_spill", 0), k.get("sgprs", 0)))
    if seen == 0:
        raise BuildError("resource report names no DP kernel: the guard did not run")
    if bad:
        raise BuildError("DP kernels must stay in registers (hand-pinned SGPRs, hand-written exec):\n  " + "\n  ".join(bad))
    return seen


def write_resource_table(kernels, path=RESOURCE_TABLE):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("kernel,total_sgprs,vgprs,agprs,scratch_bytes_per_lane,sgpr_spills,vgpr_spills,lds_bytes_static,"
                "occupancy_waves_per_simd\n")
        for k in sorted(kernels, key=lambda k: k["name"]):
            f.write('"%s",%d,%d,%d,%d,%d,%d,%d,%d\n' % (
                k["name"], k.get("sgprs", 0), k.get("vgprs", 0), k.get("agprs", 0), k.get("scratch", 0),
                k.get("sgpr_spill", 0), k.get("vgpr_spill", 0), k.get("lds", 0), k.get("occupancy", 0)))


def _compile_unit(args):
    import time
    hipcc, src, obj, extra = args
    cmd = [hipcc] + HIPCC_FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
    t0 = time.time()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return src, " ".join(cmd) + "    # %.0f s" % (time.time() - t0), p.returncode, p.stdout


def build_engine(force=False, verbose=False, extra_flags=(), lib=LIB, guard=True):
    """Compile the engine; returns the path of the shared library.
    extra_flags / lib / guard=False: instrumented builds of the tools (other name, no resource guard)."""
    if not force and lib == LIB and not needs_build():
        return lib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build", os.path.basename(lib) + ".obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = [(hipcc, src, os.path.join(objdir, obj), list(extra_flags) + flags) for src, obj, flags in UNITS]
    workers = max(1, min(len(jobs), int(os.environ.get("MAX_JOBS", "0")) or (os.cpu_count() or 1)))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        results = list(ex.map(_compile_unit, jobs))
    report = ""
    for src, cmd, rc, text in results:
        if verbose:
            print(cmd)
        if rc != 0:
            raise BuildError("hipcc failed on %s:\n" % src + "\n".join(
                l for l in text.split("\n") if "kernel-resource-usage" not in l)[-8000:])
        report += text
    kernels = parse_resource_report(report)
    n = check_resources(kernels) if guard else 0
    tmp = lib + ".tmp"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [j[2] for j in jobs] + ["-o", tmp]
    p = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise BuildError("link failed:\n" + p.stdout[-8000:])
    if lib == LIB:
        write_resource_table(kernels)
    os.replace(tmp, lib)
    if verbose:
        print("resource guard: %d DP kernel variants, none spills; table -> %s" % (n, RESOURCE_TABLE))
    return lib


HOST_JOB = os.path.join(HERE, "host", "minrms_host_job")


def build_host_job(force=False):
    """The C++ host class (host/search_or_hip.h) as a program: host/host_job.cc linked with the engine."""
    src = [os.path.join(HERE, "host", f) for f in ("host_job.cc", "search_or_hip.h")] + \
          [os.path.join(ROOT, "include", "minrms_engine.h")]
    if not force and os.path.exists(HOST_JOB) and all(os.path.getmtime(f) <= os.path.getmtime(HOST_JOB) for f in src):
        return HOST_JOB
    cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-Wall", "-Wextra", src[0], "-o", HOST_JOB,
           "-L" + HERE, "-lminrms_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise BuildError("host program failed to build:\n" + p.stdout[-4000:])
    return HOST_JOB


if __name__ == "__main__":
    print(build_engine(force=True, verbose=True))
    print(build_host_job(force=True))
