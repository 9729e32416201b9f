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


@pytest.mark.skipif(not os.path.exists(LIB), reason="libmfft.so is not built")
@pytest.mark.skipif(not all(_tool(t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")),
                    reason="ROCm LLVM tools not found")
def test_lds_dma_kernels_keep_their_counted_wait(tmp_path):
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
@pytest.mark.skipif(not all(_tool(t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")),
                    reason="ROCm LLVM tools not found")
def test_prefetching_tile_kernels_request_the_next_tile_between_exchanges(tmp_path):
    """The 512-point complex128 tile passes (fft_amd/csrc/tile_kernel.h, PF) run one workgroup per CU and
    overlap its phases by requesting the next tile into a second register set behind the first exchange.
    That holds only while the request stays between two exchanges' barriers (a __syncthreads() fence or
    a compiler that sinks the loads to their use would undo it), the barriers are LDS-only, and the
    kernels fit 256 registers without spilling inside the loop."""
    objdump = _tool("llvm-objdump")
    checked = 0
    for co in _code_objects(tmp_path):
        syms = subprocess.run([objdump, "-t", str(co)], capture_output=True, text=True, check=True).stdout
        if "fft_tile_kernelIdLi9ELi4E" not in syms:
            continue
        dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
        for name, (_, ins) in _functions(dis).items():
            m = re.search(r"fft_tile_kernelIdLi9ELi4ELi(\d)E", name)
            if not m or int(m.group(1)) not in (0, 1, 3, 5, 6, 7):    # dense input: middle, first, last, last + power, last + twiddle, first + window
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
    assert checked >= 6, f"only {checked} prefetching tile kernels found in {LIB}"
