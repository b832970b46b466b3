"""The C-ABI library loads and exports every symbol include/msnap.h declares.
No compute is launched here (these run without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    with open(os.path.join(ROOT, "include", "msnap.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msnap_[a-z0-9_]+)\s*\(", text)))


def test_library_exists_and_loads():
    from drone_path_planning_python_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    lib = _lib.load()
    assert lib.msnap_version() >= 100


def test_every_declared_symbol_is_exported():
    from drone_path_planning_python_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/msnap.h but not exported"
    # and the binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms


def test_strerror_and_argument_checks_without_gpu():
    from drone_path_planning_python_amd import _lib
    lib = _lib.load()
    assert lib.msnap_strerror(0) == b"ok"
    assert b"order" in lib.msnap_strerror(-3)
    h = ctypes.c_void_p()
    assert lib.msnap_create(ctypes.byref(h), 0, 8, 10) == -3      # MSNAP_EORDER
    assert lib.msnap_create(ctypes.byref(h), 0, 7, 0) == -4       # MSNAP_ESEGMENTS
    assert lib.msnap_create(None, 0, 7, 10) == -1                 # MSNAP_EINVAL
    rc = lib.msnap_create(ctypes.byref(h), 0, 7, 10)
    if rc == 0:       # a GPU is present: fine, release it
        lib.msnap_destroy(h)
    else:
        assert rc == -6                                           # MSNAP_ENODEVICE, no fallback
    assert lib.msnap_sync(None) == -1
    lib.msnap_destroy(None)                                       # tolerated


def test_no_cpu_fallback_in_product():
    """The package must not import anything from oracle/ (parity claims depend on it)."""
    pkg = os.path.join(ROOT, "drone_path_planning_python_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert "msnap_oracle" not in src and "c_oracle" not in src, fn
                assert "import oracle" not in src and "from oracle" not in src, fn


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    from drone_path_planning_python_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(OSError):
        _lib.load()
    monkeypatch.undo()
    importlib.reload(_lib)


def test_prefetch_wait_counts_match_the_code_object():
    """tools/check_prefetch_isa.py: the hand-counted `s_waitcnt vmcnt(N)` of the persistent solve kernel
    against the store instructions the compiler actually emitted (msnap_solve.o of this build)."""
    import subprocess
    import sys
    obj = os.path.join(ROOT, "drone_path_planning_python_amd", "csrc", "msnap_solve.o")
    if not os.path.exists(obj) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no object file / ROCm LLVM tools here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch_isa.py"), obj],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_no_register_pressure_copy_under_a_reduced_exec_mask():
    """tools/check_exec_isa.py on the objects of this build: the compiler's copies of values that do not fit the 256
    architectural VGPRs (v_accvgpr_write_b32; scratch stores) must not sit inside a divergent region -- under a reduced
    exec mask they save only some lanes.  Round 4 (and, by every sign, round 3's abort) was such a copy in the exit
    block of a loop the lanes leave one by one: the cross-tile prefetch of solve_kernel_twin<5, 20> then read at
    base + 16 * garbage."""
    import subprocess
    import sys
    objs = [os.path.join(ROOT, "drone_path_planning_python_amd", "csrc", f) for f in ("msnap_solve.o", "msnap_aux.o", "msnap_grid.o")]
    if not all(os.path.exists(o) for o in objs) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no object files / ROCm LLVM tools here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_exec_isa.py")] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("none under a reduced exec mask") == 3, r.stdout


def test_the_exec_check_sees_the_round_4_fault_pattern():
    """tools/check_exec_isa.walk on the instruction shapes of the faulty build (solve_kernel_twin<5, 20>, round 4): a
    loop the lanes leave one by one, the copies in its exit block in front of the restore -- reported; the same copies
    behind the restore, and copies around a wave-uniform loop with a predicated body -- not reported."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_exec_isa as chk

    def body(lines):
        out, labels = [], {}
        for ln in lines:
            if ln.endswith(":"):
                labels[ln[:-1]] = 0x1000 + 4 * len(out)
            else:
                out.append(ln)
        res = []
        for i, ln in enumerate(out):
            ins, _, lab = ln.partition(" -> ")
            res.append((ins, 0x1000 + 4 * i, labels[lab] if lab else None))
        return res

    faulty = body(["v_add_u32_e32 v22, 64, v28",
                   "s_and_saveexec_b64 s[4:5], s[0:1]",
                   "s_cbranch_execz 58 -> join",
                   "loop:",
                   "global_store_dwordx2 v[0:1], v[4:5], off",
                   "s_or_b64 s[6:7], s[0:1], s[6:7]",
                   "s_andn2_b64 exec, exec, s[6:7]",
                   "s_cbranch_execnz 65501 -> loop",
                   "v_accvgpr_write_b32 a4, v18",
                   "v_accvgpr_write_b32 a2, v22",
                   "join:",
                   "s_or_b64 exec, exec, s[4:5]",
                   "s_endpgm"])
    found = chk.walk(faulty)
    assert sorted(f[0] for f in found) == ["v_accvgpr_write_b32 a2, v22", "v_accvgpr_write_b32 a4, v18"], found
    behind = body(faulty_lines := ["s_and_saveexec_b64 s[4:5], s[0:1]", "s_cbranch_execz 3 -> join",
                                   "global_store_dwordx2 v[0:1], v[4:5], off", "join:", "s_or_b64 exec, exec, s[4:5]",
                                   "v_accvgpr_write_b32 a4, v18", "s_endpgm"])
    assert chk.walk(behind) == [], faulty_lines
    # the same placement with an ordinary register move (what a kernel without accumulation registers would get): behind
    # the latch of a loop that runs until no lane is left, exec is zero on every execution -- any vector instruction there
    # is reported
    moved = body([ln.replace("v_accvgpr_write_b32 a4, v18", "v_mov_b32_e32 v40, v18").replace("v_accvgpr_write_b32 a2, v22", "s_nop 0")
                  for ln in ["s_and_saveexec_b64 s[4:5], s[0:1]", "s_cbranch_execz 58 -> join", "loop:",
                             "global_store_dwordx2 v[0:1], v[4:5], off", "s_andn2_b64 exec, exec, s[6:7]",
                             "s_cbranch_execnz 65501 -> loop", "v_accvgpr_write_b32 a4, v18", "v_accvgpr_write_b32 a2, v22",
                             "join:", "s_or_b64 exec, exec, s[4:5]", "v_mov_b32_e32 v41, v18", "s_endpgm"]])
    assert [f[0] for f in chk.walk(moved)] == ["v_mov_b32_e32 v40, v18"]
    uniform = body(["loop:",
                    "s_and_saveexec_b64 s[4:5], vcc",
                    "s_xor_b64 s[0:1], exec, s[4:5]",          # the saved mask changes its register
                    "global_store_dwordx2 v[0:1], v[4:5], off",
                    "s_or_b64 exec, exec, s[0:1]",
                    "s_cmp_lt_i32 s8, s9",
                    "s_cbranch_scc1 65500 -> loop",
                    "v_accvgpr_write_b32 a4, v18",
                    "s_endpgm"])
    assert chk.walk(uniform) == []


def test_no_kernel_of_the_library_uses_scratch():
    """Code-object metadata of every kernel in libmsnap.so: .private_segment_fixed_size == 0.  Spilled registers cost
    memory round trips inside the hot loops, and scratch ties a launch to per-queue state of the runtime (an order-9
    instance with 188 bytes of scratch aborted inside the HIP runtime when a long-lived stream first needed scratch
    after short-lived ones had used it): the kernels are built to their register budgets instead."""
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(ROOT, "drone_path_planning_python_amd", "csrc", "libmsnap.so")
    if not os.path.exists(lib) or not os.path.exists(f"{llvm}/llvm-readelf"):
        pytest.skip("no library / ROCm LLVM tools here")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "lib.co")
        subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")],
                       check=True)
        # libmsnap.so bundles one code object per translation unit
        raw = open(fat, "rb").read()
        names, sizes = [], []
        off = 0
        while True:
            k = raw.find(b"__CLANG_OFFLOAD_BUNDLE__", off)
            if k < 0:
                break
            piece = os.path.join(tmp, f"bundle{len(names)}.bin")
            nxt = raw.find(b"__CLANG_OFFLOAD_BUNDLE__", k + 8)
            open(piece, "wb").write(raw[k:nxt if nxt > 0 else len(raw)])
            r = subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={piece}", f"--output={co}"],
                               capture_output=True, text=True)
            off = k + 8
            if r.returncode != 0:
                continue
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            names += re.findall(r"\.name:\s+(\S+)", notes)
            sizes += [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
    assert len(sizes) >= 100 and len(sizes) == len(names), (len(sizes), len(names))
    bad = [(n, s) for n, s in zip(names, sizes) if s != 0]
    assert not bad, bad[:5]
