"""Build the in-tree HIP library polar_amd/libpolar_amd.so for gfx950.

hipcc cross-compiles without a GPU. The shared object is linked against the libamdhip64 that
PyTorch-ROCm bundles (DT_NEEDED "libamdhip64.so"), so that when it is loaded into a process
that already imported torch both share ONE HIP runtime (device pointers and streams handed
over from torch are then valid); stand-alone (C++/MATLAB hosts) the same name resolves to
/opt/rocm/lib through the rpath.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libpolar_amd.so")
BUILD = os.path.join(HERE, "_build")
# (source, extra -D, object tag, extra compiler options): polar_kernels.hip is compiled four times — LLR-domain kernel family,
# exp-domain kernels of the small groups, exp-domain list of 32 (the headline kernel, with its own scheduler options), exp-domain
# one-codeword-per-wave kernels
FLAGS_LIST32 = ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause", "-mllvm", "-amdgpu-use-amdgpu-trackers"] + \
               os.environ.get("POLAR_LIST32_FLAGS", "").split()          # (A/B experiments on the list-of-32 translation unit alone)
# the one-codeword-per-wave kernels (a lone wave: every instruction is 4 cycles, a taken branch 20 — DESIGN.md §7)
FLAGS_LAT = os.environ.get("POLAR_LAT_FLAGS", "").split()
SOURCES = [("polar_kernels.hip", ["POLAR_ED_TU=0"], "", []), ("polar_kernels.hip", ["POLAR_ED_TU=1"], ".ed", []),
           ("polar_kernels.hip", ["POLAR_ED_TU=2"], ".ed32", FLAGS_LIST32), ("polar_kernels.hip", ["POLAR_ED_TU=3"], ".lat", FLAGS_LAT),
           ("polar_kernels_sc.hip", [], "", []), ("polar_kernels_p1.hip", [], "", []), ("polar_channel.hip", [], "", []),
           ("polar_construct.hip", [], "", [])] + \
          [(f, [], "", []) for f in ("polar_handle.cpp", "polar_decode.cpp", "polar_hostpipe.cpp", "polar_montecarlo.cpp", "polar_multi.cpp",
                                     "polar_debug.cpp")]
# the test build (libpolar_amd_test.so): the same objects, except that these two are compiled again with -DPOLAR_TEST_HOOKS —
# the fault-injection keys of include/polar_amd_debug.h exist only there
TEST_HOOK_SOURCES = ("polar_montecarlo.cpp", "polar_debug.cpp")
LIB_TEST = os.path.join(HERE, "libpolar_amd_test.so")
# what POLAR_DEFS may name: instrumentation (measures, does not decode), trimmed development builds (fewer instantiations),
# and the few A/B switches still open (same bits either way)
KNOWN_DEFS = {"POLAR_MARGIN", "POLAR_SLOTHIST", "SCLAT_PROF", "POLAR_DEV_GS32", "POLAR_DEV_ONE", "POLAR_NO_FIXED_N",
              "POLAR_NO_GUARD_INT", "POLAR_NO_COLD_HINTS", "OCC", "ED_NR"}
ARCH = os.environ.get("POLAR_ARCH", "gfx950")      # (A/B: e.g. gfx950:xnack-)
_flags_ok = {}


def _probe_flags(flags):
    """The scheduler options of the list-of-32 translation unit are hidden LLVM options (-mllvm ...): a ROCm / LLVM that
    does not know one of them stops with 'Unknown command line argument' and the whole build fails. Probed once per
    process on an empty translation unit; unknown options are dropped (the kernel is then built with the default scheduler:
    a few per cent slower, same results) and the fact is printed."""
    key = tuple(flags)
    if key not in _flags_ok:
        import tempfile
        ok = list(flags)
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            open(src, "w").write("#include <hip/hip_runtime.h>\n__global__ void probe_kernel() {}\n")
            pairs = [flags[i:i + 2] for i in range(0, len(flags), 2)]          # ("-mllvm", "<option>")
            ok = []
            for pr in pairs:
                r = subprocess.run([_hipcc(), "--offload-arch=" + ARCH, "-c", src, "-o", os.path.join(d, "probe.o")] + pr,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
                if r.returncode == 0 or "Unknown command line argument" not in r.stderr:
                    ok += pr              # (a probe that fails for any OTHER reason decides nothing: keep the option)
                else:
                    print("polar_amd.build: this compiler rejects %s (%s); building without it" %
                          (" ".join(pr), (r.stderr.strip().splitlines() or ["?"])[-1][:120]), file=sys.stderr)
        _flags_ok[key] = ok
    return _flags_ok[key]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _torch_lib():
    try:
        import torch  # noqa: F401  (plumbing only: locate the bundled HIP runtime)
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    except Exception:
        pass
    return None


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _fingerprint(cmd, files):
    """sha256 of the compile command and of the CONTENT of the source and every header it includes: an object is rebuilt when
    this changes, not when a timestamp does (a checkout or a copy touches files without changing them, and a rebuilt library is
    not byte-identical — the committed rocprofv3 profiles are keyed by the library's hash, bench.py). Paths enter relative to
    the repository root: the tree is copied to another place on the GPU box and must not rebuild there."""
    import hashlib
    root = os.path.realpath(ROOT)

    def rel(x):
        rx = os.path.realpath(x) if os.path.isabs(x) else x
        return "<root>" + rx[len(root):] if os.path.isabs(rx) and rx.startswith(root) else x
    h = hashlib.sha256(" ".join(rel(c) for c in cmd[1:]).encode())     # (not the compiler's own path)
    for f in sorted(set(files), key=rel):
        try:
            h.update(rel(f).encode()); h.update(open(f, "rb").read())
        except OSError:
            h.update(b"<missing>")
    return h.hexdigest()


def _deps(obj, fallback):
    """Headers an object really includes (the compiler's -MD file next to it); every header when there is none yet."""
    d = obj + ".d"
    if not os.path.exists(d):
        return fallback
    txt = open(d).read().replace("\\\n", " ")
    parts = txt.split(":", 1)
    if len(parts) < 2:
        return fallback
    # (the .d file holds the absolute paths of the place the object was compiled in; the tree may have been copied since:
    # re-anchor every in-tree header at THIS root by its path below polar_amd/ or include/)
    out = []
    for x in parts[1].split():
        for anchor in ("/polar_amd/csrc/", "/include/"):
            k = x.rfind(anchor)
            if k >= 0 and not x.startswith("/opt/") and not x.startswith("/usr/"):
                cand = os.path.join(ROOT, x[k + 1:])
                if os.path.exists(cand):
                    out.append(cand)
                break
    return out


def build(force=False, verbose=False, profile=False, bless=False, test_hooks=False):
    """profile=True builds the instrumented variant libpolar_amd_prof.so (-DPOLAR_PROFILE: per-phase
    cycle counters, tools/phase_profile.py); never used by the product path.
    test_hooks=True builds libpolar_amd_test.so: the product's objects, except TEST_HOOK_SOURCES compiled with
    -DPOLAR_TEST_HOOKS (the fault-injection keys of include/polar_amd_debug.h) — what the tests of the failure protocol load."""
    global LIB
    os.makedirs(BUILD, exist_ok=True)
    lib_out = os.path.join(HERE, "libpolar_amd_prof.so") if profile else (LIB_TEST if test_hooks else LIB)
    tag = ".prof" if profile else ""
    if os.environ.get("POLAR_BUILD_TAG"):          # A/B experiments: separate objects and library
        tag = "." + os.environ["POLAR_BUILD_TAG"]
        lib_out = os.path.join(HERE, "libpolar_amd_%s%s.so" % (os.environ["POLAR_BUILD_TAG"], "_test" if test_hooks else ""))
    # Extra -D macros (POLAR_DEFS) select instrumented or trimmed builds — some of them measure instead of decoding (POLAR_MARGIN
    # overwrites decoded bits with its statistics). They never get the product's library name: a tag is required, every name must be
    # one this tree knows (a typo would otherwise build an ordinary library under an experiment's name, or the reverse), and
    # POLAR_DEV_BUILD is defined for them (csrc/polar_device.h stops an instrumented translation unit that lacks it).
    extra_defs = os.environ.get("POLAR_DEFS", "").split()
    dev_build = profile or bool(extra_defs)
    if extra_defs:
        unknown = [d for d in extra_defs if d.split("=")[0] not in KNOWN_DEFS]
        if unknown:
            raise SystemExit("polar_amd.build: POLAR_DEFS names unknown macro(s) %s (known: %s)" % (unknown, sorted(KNOWN_DEFS)))
        if not os.environ.get("POLAR_BUILD_TAG"):
            raise SystemExit("polar_amd.build: POLAR_DEFS=%r without POLAR_BUILD_TAG: development builds never take the product "
                             "library's name (libpolar_amd.so)" % os.environ["POLAR_DEFS"])
    headers = [os.path.join(INC, f) for f in os.listdir(INC)] + \
              [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    objs = []
    jobs = []
    for s, defs, otag, xflags in SOURCES:
        src = os.path.join(CSRC, s)
        if test_hooks and s in TEST_HOOK_SOURCES:
            defs, otag = defs + ["POLAR_TEST_HOOKS"], otag + ".th"
        obj = os.path.join(BUILD, s + otag + tag + ".o")
        objs.append(obj)
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-x", "hip", "-I", INC, "-I", CSRC, "-MD", "-MF", obj + ".d",
               "-c", src, "-o", obj]
        if profile:
            cmd.insert(1, "-DPOLAR_PROFILE")
        for d in defs + extra_defs + (["POLAR_DEV_BUILD"] if dev_build else []):
            cmd.insert(1, "-D" + d)
        cmd[1:1] = (_probe_flags(xflags) if xflags else []) + os.environ.get("POLAR_HIPCC_FLAGS", "").split()    # (+ A/B experiments with compiler options)
        fp = _fingerprint(cmd, [src] + _deps(obj, headers))
        stamp = obj + ".sha"
        have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(obj) else None
        if bless and os.path.exists(obj):
            # DEVELOPMENT ONLY: adopt the existing object as built from the current sources (objects that predate the
            # stamps). Nothing checks that it really was: a stale object shipped this way no longer matches the sources.
            if have != fp:
                print("polar_amd.build --bless: WARNING: stamping %s as current WITHOUT compiling it; use only on a tree "
                      "whose objects you know to be built from these sources" % os.path.basename(obj), file=sys.stderr)
            open(stamp, "w").write(fp)
            continue
        if force or have != fp:
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, stamp, obj, src, headers))
    if jobs:   # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(job):
            cmd, stamp, obj, src, hdrs = job
            subprocess.check_call(cmd)
            open(stamp, "w").write(_fingerprint(cmd, [src] + _deps(obj, hdrs)))     # (with the dependency list the compiler just wrote)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    # (the link, too, is decided by content: the stamps of the objects it was made from)
    import hashlib
    link_fp = hashlib.sha256("".join(open(o + ".sha").read() if os.path.exists(o + ".sha") else "?" for o in objs).encode()).hexdigest()
    link_stamp = os.path.join(BUILD, os.path.basename(lib_out) + ".sha")
    have_link = open(link_stamp).read().strip() if os.path.exists(link_stamp) and os.path.exists(lib_out) else None
    if bless and os.path.exists(lib_out):
        open(link_stamp, "w").write(link_fp)
        have_link = link_fp
    if force or jobs or have_link != link_fp:
        tl = _torch_lib()
        # only the C entry points of include/polar_amd*.h leave the library (kernel launchers and internals stay local)
        vs = os.path.join(BUILD, "exports.map")
        open(vs, "w").write("{ global: polar_*; local: *; };\n")
        cmd = [_hipcc(), "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", lib_out] + objs
        if tl:
            # hipcc would add -L/opt/rocm/lib -lamdhip64 itself; link by hand instead so the
            # DT_NEEDED entry is the un-versioned name torch's copy is loaded under
            clang = "/opt/rocm/lib/llvm/bin/clang++"
            cmd = [clang, "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", lib_out] + objs + \
                  ["-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl, "-Wl,-rpath,/opt/rocm/lib", "-lstdc++", "-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        open(link_stamp, "w").write(link_fp)
    return lib_out


def build_test(force=False):
    """libpolar_amd_test.so (see build)."""
    return build(force=force, test_hooks=True)


def build_cli(force=False):
    """C++ driver mirroring the reference's main.cpp over the header-only class mirror
    (polar_amd/cpp/PolarCode.hpp -> C-ABI)."""
    lib = build()
    exe = os.path.join(BUILD, "polar_main")
    src = os.path.join(HERE, "cpp", "main.cpp")
    hdr = os.path.join(HERE, "cpp", "PolarCode.hpp")
    if force or _newer(exe, [src, hdr, lib]):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", INC, "-I", os.path.join(HERE, "cpp"), src, "-o", exe,
                               "-L", HERE, "-lpolar_amd", "-Wl,-rpath," + HERE,
                               "-Wl,-rpath," + (_torch_lib() or "/opt/rocm/lib"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


if __name__ == "__main__":
    if "--bless" in sys.argv and not os.environ.get("POLAR_DEV"):
        raise SystemExit("polar_amd.build: --bless stamps existing objects as current without compiling them; it is a "
                         "development aid and needs POLAR_DEV=1 in the environment")
    print(build(force="--force" in sys.argv, verbose=True, profile="--profile" in sys.argv, bless="--bless" in sys.argv))
