"""Invariants of the compiled LDS-DMA kernels, checked on the machine code inside libmfft.so.

The pipeline of those kernels (fft_amd/csrc/lds_kernel.h) ends every trip with a COUNTED wait,
`s_waitcnt vmcnt(P)`: the next signal's LDS-DMA loads were issued before the trip's P stores, vector
memory operations retire in issue order, so "at most P outstanding" means the DMA has landed while the
stores keep draining.  That is only true while the compiler emits at least P store instructions
between the DMA and the wait, and it only pays while nothing else inside the loop waits on the counter
(a reload from scratch, a table read, a __syncthreads() fence).  A compiler upgrade or an innocent
edit can break either silently -- the first one into wrong results on some launches -- so the
disassembly is checked here.  CPU-only: needs the ROCm LLVM tools, not a GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fft_amd", "lib", "libmfft.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def _code_objects(tmp_path):
    """gfx950 code objects embedded in the library, one per translation unit."""
    objcopy, bundler = _tool("llvm-objcopy"), _tool("clang-offload-bundler")
    fat = tmp_path / "fat.bin"
    subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", LIB, str(tmp_path / "unused.so")], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for k, s in enumerate(starts):
        part = tmp_path / f"bundle{k}.bin"
        part.write_bytes(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = tmp_path / f"co{k}.o"
        r = subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={part}", f"--output={co}"], capture_output=True)
        if r.returncode == 0 and co.exists() and co.stat().st_size > 0:
            out.append(co)
    return out


def _functions(disasm):
    """{symbol: (start address, [(address, instruction text, branch target or None)])} of an
    llvm-objdump -d listing (addresses from the trailing comments, targets from `<symbol+0x..>`)."""
    funcs, cur, start = {}, None, 0
    for line in disasm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            start = int(m.group(1), 16)
            cur = funcs.setdefault(m.group(2), (start, []))[1]
            continue
        if cur is None or not line.startswith("\t"):
            continue
        am = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        if not am:
            continue
        text = line.strip().split("//")[0].strip()
        tm = re.search(r"<\S+?\+0x([0-9a-fA-F]+)>\s*$", line)
        target = start + int(tm.group(1), 16) if (tm and text.startswith(("s_branch", "s_cbranch"))) else None
        cur.append((int(am.group(1), 16), text, target))
    return funcs


def _need_tools():
    """The checks below guard results, not speed: where the library was built the ROCm LLVM tools exist, so their
    absence is a failure, not a reason to skip (MFFT_ALLOW_NO_LLVM_TOOLS=1 turns it into a skip on purpose)."""
    missing = [t for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump") if not _tool(t)]
    if missing:
        if os.environ.get("MFFT_ALLOW_NO_LLVM_TOOLS") == "1":
            pytest.skip(f"ROCm LLVM tools not found: {missing}")
        pytest.fail(f"ROCm LLVM tools not found ({missing}): the machine-code invariants of libmfft.so cannot be checked")


@pytest.mark.skipif(not os.path.exists(LIB), reason="libmfft.so is not built")
def test_lds_dma_kernels_keep_their_counted_wait(tmp_path):
    _need_tools()
    objdump = _tool("llvm-objdump")
    checked = 0
    for co in _code_objects(tmp_path):
        syms = subprocess.run([objdump, "-t", str(co)], capture_output=True, text=True, check=True).stdout
        if "fft_lds_kernel" not in syms:
            continue
        dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
        for name, (_, ins) in _functions(dis).items():
            if "fft_lds_kernel" not in name or not any("global_load_lds_dwordx4" in t for _, t, _ in ins):
                continue
            checked += 1
            # the batch loop = the backward branch with the longest span (table-fill and waterfall loops are short);
            # its blocks occupy the address range [target, branch]
            back = [(adr - tgt, tgt, adr) for adr, _, tgt in ins if tgt is not None and tgt < adr]
            assert back, name
            _, lo, hi = max(back)
            body = [t for adr, t, _ in ins if lo <= adr <= hi]
            assert any("global_load_lds_dwordx4" in t for t in body), (name, "the loop found holds no DMA")
            waits = [int(m.group(1)) for t in body for m in [re.search(r"vmcnt\((\d+)\)", t)] if m]
            assert len(waits) == 1, (name, "waits on the vector-memory counter inside the loop", waits)
            left = waits[0]
            assert left > 0, (name, "the loop's wait drains the stores")
            stores = sum(1 for t in body if re.match(r"(buffer|global)_store_", t))
            assert stores >= left, (name, f"vmcnt({left}) behind {stores} stores would let DMA loads be outstanding")
            counted = [t for t in body if re.match(r"(global_load_dword|buffer_load|scratch_load|flat_load)", t)]
            assert not counted, (name, "compiler-counted loads inside the loop", counted[:4])
            # a descriptor the compiler cannot prove uniform costs a waterfall loop per store
            assert not any(t.startswith("v_readfirstlane") for t in body[len(body) // 2:]), (name, "waterfall loop around a store")
    assert checked >= 8, f"only {checked} LDS-DMA kernels found in {LIB}"


@pytest.mark.skipif(not os.path.exists(LIB), reason="libmfft.so is not built")
def test_prefetching_tile_kernels_request_the_next_tile_between_exchanges(tmp_path):
    """The 512-point complex128 tile passes (fft_amd/csrc/tile_kernel.h, PF) run one workgroup per CU and
    overlap its phases by requesting the next tile into a second register set behind the first exchange;
    so do the 1024 x 8 and 2048 x 4 ones of the three-pass plans.  That holds only while the request stays between two exchanges' barriers (a __syncthreads() fence or
    a compiler that sinks the loads to their use would undo it), the barriers are LDS-only, and the
    kernels fit 256 registers without spilling inside the loop."""
    _need_tools()
    objdump = _tool("llvm-objdump")
    checked = 0
    for co in _code_objects(tmp_path):
        syms = subprocess.run([objdump, "-t", str(co)], capture_output=True, text=True, check=True).stdout
        if "fft_tile_kernelIdLi9ELi4E" not in syms:
            continue
        dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
        for name, (_, ins) in _functions(dis).items():
            m = re.search(r"fft_tile_kernelId(Li9ELi4|Li10ELi3|Li11ELi2)ELi(\d)E", name)
            # dense input: middle, first, last, last + power, last + twiddle; the 512-point tiles also first + window
            if not m or int(m.group(2)) not in ((0, 1, 3, 5, 6, 7) if m.group(1) == "Li9ELi4" else (0, 1, 3, 5, 6)):
                continue
            checked += 1
            back = [(adr - tgt, tgt, adr) for adr, _, tgt in ins if tgt is not None and tgt < adr]
            assert back, name
            _, lo, hi = max(back)
            body = [t for adr, t, _ in ins if lo <= adr <= hi]
            assert not any(t.startswith(("scratch_load", "scratch_store")) for t in body), (name, "spills inside the loop")
            # positions of barriers and of sample loads: some load must have a barrier before AND after it
            bars = [i for i, t in enumerate(body) if t.startswith("s_barrier")]
            loads = [i for i, t in enumerate(body) if t.startswith("global_load_dwordx4")]
            assert len(bars) >= 8 and loads, (name, len(bars), len(loads))    # two trips x two exchanges x two barriers
            assert any(bars[0] < i < bars[-1] for i in loads), (name, "the next tile is not requested between the exchanges")
            # a full fence before a barrier would drain the request
            for b in bars:
                window = " ".join(body[max(0, b - 3):b])
                assert "vmcnt(0)" not in window, (name, "a barrier of the exchange drains the vector-memory counter")
    assert checked >= 16, f"only {checked} prefetching tile kernels found in {LIB}"


# Non-default values of the kernels' MFFT_* build macros (A/B builds: make VARIANT=x KNOBS="-D...").  The default
# build never takes these branches, so they are type-checked here: three syntax-only device compiles of a
# translation unit that instantiates every variant of a few sizes of each kernel family (a few seconds each).
KNOB_SETS = [
    "-DMFFT_TW_SRC=2 -DMFFT_WIN_SRC=2 -DMFFT_PREFETCH=3 -DMFFT_AFFINE=1 -DMFFT_VEC=1 -DMFFT_FAST_BUF=1 -DMFFT_TILE_EVAL_ROOTS=1 "
    "-DMFFT_BIG_RADIX=1 -DMFFT_TILE_PF_GEN=1 -DMFFT_P8192=16 -DMFFT_WAVES_PER_EU=2 -DMFFT_BLOCK_MIN=128 -DMFFT_BLOCK_1024_HANN=256 -DMFFT_WIN_BUF_FROM=4096 "
    "-DMFFT_WAVE_UNROLL=2",
    "-DMFFT_TW_SRC=0 -DMFFT_PREFETCH=0 -DMFFT_AFFINE=0 -DMFFT_VEC=0 -DMFFT_NT=0 -DMFFT_DMA=0 -DMFFT_TILE_TW2=0 -DMFFT_TILE_VEC=0 "
    "-DMFFT_TILE_EVAL_ROOTS=0 -DMFFT_TILE_PREFETCH=0 -DMFFT_TILE_OCC=0 -DMFFT_STATS_F32=0 -DMFFT_STATS_CLOSED=0 "
    "-DMFFT_DMA_STATS_LDS=0 -DMFFT_DMA_WARM=0 -DMFFT_BLOCK_2048=0 -DMFFT_C128_1024_COLS=2 -DMFFT_TILE_PF_HANN=0 -DMFFT_TILE_PF_POWER=0",
    "-DMFFT_DIAG_NOFFT=1 -DMFFT_DIAG_SKIP=3 -DMFFT_DIAG_STAMPS=1",
]


@pytest.mark.parametrize("knobs", KNOB_SETS)
def test_build_knobs_still_compile(knobs):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    src = os.path.join(ROOT, "tools", "csrc", "knob_check.hip")
    r = subprocess.run([hipcc, "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-fsyntax-only",
                        "-Wno-unused-function", *knobs.split(), src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # every macro the kernels' headers define as a knob is named by one of the sets (a new knob needs a line here)
    defined = set()
    for h in ("lds_kernel.h", "tile_kernel.h", "wave_kernel.h"):
        with open(os.path.join(ROOT, "fft_amd", "csrc", h)) as fh:
            defined |= set(re.findall(r"^#ifndef (MFFT_[A-Z0-9_]+)\n#define \1 ", fh.read(), re.M))
    named = set(re.findall(r"-D(MFFT_[A-Z0-9_]+)=", " ".join(KNOB_SETS)))
    assert defined - named <= {"MFFT_R1024"}, sorted(defined - named)
