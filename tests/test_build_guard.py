"""Build-time guard for the hand-scheduled kernels.

k_conv16 / k_conv16b / k_convlat stage their operands with inline-asm LDS-DMA (`global_load_lds_dwordx4`) and
wait for them with hand-counted `s_waitcnt vmcnt(N)` immediates; the asm overwrites m0 / exec around each
DMA without declaring it.  Those counts are only right while the compiler issues NO vector-memory operation of
its own between a DMA and its wait -- a scratch spill (or a new compiler that schedules differently) would turn
the waits into a silent LDS race: wrong bits in rare tiles, encoder and decoder diverging.  The bit-exact GPU
tests (test_gpu_fullsize.py: every LDS-DMA kernel against the register-staged general kernel at full size,
test_small_grid_kernel_agrees_with_k_conv16, test_soak_lds_dma_kernels) catch that on hardware; this test
catches its usual cause on the CPU, from the code object hipcc just built: none of these kernels may use
scratch memory or spill a register, and the compiler must be the one the counts were validated with."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "tezip_amd", "csrc", "tz_prednet.o")
KNOWN_GOOD_COMPILERS = ("roc-7.2.0",)   # hipcc --version of the toolchains the vmcnt immediates were validated with


def _kernel_metadata(tmp_path):
    if not os.path.exists(OBJ):
        from tezip_amd import build
        build.build()
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(OBJ, work / "k.o")   # llvm-objdump --offloading writes the bundles next to its input
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "k.o"], cwd=work, stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(work) if "amdgcn" in f]
    assert len(co) == 1, co
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co[0]], cwd=work, text=True)
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s+\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "name" and val.startswith("_Z"):
            cur = kernels.setdefault(val, {})
        elif cur is not None and key in ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "vgpr_count"):
            cur[key] = int(val)
    return kernels


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no ROCm LLVM tools")
def test_hand_scheduled_kernels_use_no_scratch_and_spill_nothing(tmp_path):
    kernels = _kernel_metadata(tmp_path)
    guarded = {n: k for n, k in kernels.items() if re.search(r"k_convlat|k_conv16", n)}
    assert len(guarded) >= 12, sorted(kernels)   # k_conv16 / k_conv16b / k_convlat / k_convlat_pair instantiations
    for name, k in guarded.items():
        assert k.get("private_segment_fixed_size") == 0, (name, k)   # scratch = compiler-issued VMEM between DMA and wait
        assert k.get("sgpr_spill_count") == 0 and k.get("vgpr_spill_count") == 0, (name, k)
        assert k.get("vgpr_count", 999) <= 128, (name, k)            # the launch bounds these kernels were tuned for


def test_compiler_is_the_one_the_waitcnt_immediates_were_validated_with():
    out = subprocess.check_output([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], text=True)
    assert any(tag in out for tag in KNOWN_GOOD_COMPILERS), (
        "hipcc changed:\n%s\nre-run the bit-exact GPU tests (pytest -m gpu tests/test_gpu_fullsize.py) and scripts/soak_conv.py "
        "with this compiler, then add its tag to KNOWN_GOOD_COMPILERS" % out)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no ROCm LLVM tools")
def test_k_wino_stage_bodies_move_no_register_behind_an_asm_read(tmp_path):
    """k_wino (TZ-PA2, tz_wino_kernels.hip.h) reads LDS through asm the compiler does not see as loads and multiplies with asm
    MFMAs on a tied AGPR accumulator.  What nobody checks for us: between an asm read and the counted wait in front of its use
    the compiler may copy or reuse the destination register -- the data then lands in a register that means something else
    (round 4 met exactly that with asm GLOBAL loads hoisted over the output transform: a late write into an address register,
    a memory fault; that code is gone).  The static half of the guard: in the code object just built, the straight-line stage
    bodies (the blocks with 16 or more MFMAs) of every k_wino instantiation contain no register move, no AGPR <-> VGPR
    traffic, no lane spill and no scratch access at all.  The dynamic half: tests/test_gpu_wino.py and scripts/soak_conv.py
    (bit for bit against the oracle and against the plain k_wino_ref)."""
    if not os.path.exists(OBJ):
        from tezip_amd import build
        build.build()
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(OBJ, work / "k.o")
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "k.o"], cwd=work, stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(work) if "amdgcn" in f]
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co[0]], cwd=work, text=True)
    seen = 0
    for fn in re.split(r"\n(?=[0-9a-f]+ <)", text):
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", fn)
        if not m or "k_wino" not in m.group(1) or "k_wino_ref" in m.group(1):
            continue
        seen += 1
        blocks = [[]]
        for line in fn.splitlines()[1:]:
            ins = line.split("//")[0].strip()
            blocks[-1].append(ins)
            if ins.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                blocks.append([])
        hot = [b for b in blocks if sum(i.startswith("v_mfma") for i in b) >= 16]
        assert len(hot) >= 2, (m.group(1), len(hot))
        for b in hot:
            bad = [i for i in b if i.startswith(("v_mov_b32", "v_pk_mov", "v_accvgpr", "scratch_", "buffer_", "v_readlane", "v_writelane"))]
            assert not bad, (m.group(1), bad[:8])
            assert all("a[" in i.split(",")[0] for i in b if i.startswith("v_mfma")), m.group(1)   # accumulators in AGPRs, in place
    assert seen >= 6   # LSTM / RAW with and without an upsampled source, pool + error with 3 and 4 column tiles


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no ROCm LLVM tools")
def test_k_wino_reads_no_accumulator_in_the_shadow_of_an_mfma(tmp_path):
    """The other half of the asm-MFMA contract (the guard above covers the stage bodies only): the compiler does
    not know that the asm MFMAs write the AGPRs it reads in the output transform and in the epilogues, so it inserts no
    wait states for them -- the kernel does, by hand (`s_nop 15` twice behind each stage loop; a v_mfma_f32_16x16x4_f32
    needs 8 passes = up to 18 wait states before another unit may read its result).  Statically, in the code object just
    built: walking back from EVERY instruction of a k_wino function that reads an AGPR and is not itself an MFMA, at least
    18 wait states of instructions (s_nop N counts N + 1) lie between it and the nearest MFMA in program order."""
    if not os.path.exists(OBJ):
        from tezip_amd import build
        build.build()
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(OBJ, work / "k.o")
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "k.o"], cwd=work, stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(work) if "amdgcn" in f]
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co[0]], cwd=work, text=True)
    NEED = 18
    seen = readers = 0
    for fn in re.split(r"\n(?=[0-9a-f]+ <)", text):
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", fn)
        if not m or "k_wino" not in m.group(1) or "k_wino_ref" in m.group(1):
            continue
        seen += 1
        body = [l.split("//")[0].strip() for l in fn.splitlines()[1:] if l.strip()]
        for i, ins in enumerate(body):
            if ins.startswith("v_mfma") or not re.search(r"\ba(\d+|\[\d+:\d+\])", ins.split(" ", 1)[1] if " " in ins else ""):
                continue
            ops = ins.split(" ", 1)[1].split(",")
            srcs = ops[1:] if ins.startswith(("v_accvgpr_read", "v_")) else ops   # (stores / ds_write: every operand is a source)
            if not any(re.search(r"\ba(\d+|\[\d+:\d+\])", o) for o in srcs):
                continue
            readers += 1
            waited, j = 0, i - 1
            while j >= 0 and waited < NEED:
                prev = body[j]
                if prev.startswith(("s_branch", "s_endpgm")):   # nothing falls through an unconditional branch: this block is
                    break                                        # entered by a jump (a loop that runs zero times skips its MFMAs)
                assert not prev.startswith("v_mfma"), (m.group(1), "accumulator read %d wait states behind an MFMA" % waited, body[j:i + 1][:8])
                nop = re.match(r"s_nop (\d+)", prev)
                waited += int(nop.group(1)) + 1 if nop else 1
                j -= 1
    assert seen >= 7 and readers >= 7 * 96   # every instantiation reads its accumulators somewhere


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no ROCm LLVM tools")
def test_k_sse_decide_performs_its_partial_before_it_draws_its_ticket(tmp_path):
    """k_sse_decide (tz_api.hip; the DWP decision of compress.py:245-264 in one launch): every workgroup publishes its
    block's partial sum and then takes a ticket; the workgroup with the last ticket adds the partials.  The partial must
    have been PERFORMED before the ticket increment is issued -- two relaxed atomics on different addresses are not
    ordered by issue order.  Round 5 expressed that with `1u + (old & 0)`, which the compiler folded: the object code had
    a non-returning swap and no wait in front of the add.  Read off the code object just built:
    the exchange is the returning form (`sc0`), an `s_waitcnt vmcnt(0)` follows it with no other vector-memory
    instruction and no branch in between, and only then comes the ticket's `global_atomic_add ... sc0`; the reader's
    loads of part[] are agent-scope atomic loads (`global_load_dwordx2 ... sc1`, which bypass the CU's L1; an atomic
    exchange does not leave its line in the writer's L2) and all come behind that add."""
    obj = os.path.join(ROOT, "tezip_amd", "csrc", "tz_api.o")
    if not os.path.exists(obj):
        from tezip_amd import build
        build.build()
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(obj, work / "k.o")
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "k.o"], cwd=work, stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(work) if "amdgcn" in f]
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co[0]], cwd=work, text=True)
    body = None
    for fn in re.split(r"\n(?=[0-9a-f]+ <)", text):
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", fn)
        if m and "k_sse_decide" in m.group(1):
            assert body is None, "two k_sse_decide functions"
            body = [l.split("//")[0].strip() for l in fn.splitlines()[1:] if l.strip()]
    assert body, "k_sse_decide not in tz_api.o"
    swaps = [i for i, ins in enumerate(body) if ins.startswith("global_atomic_swap")]
    assert len(swaps) == 1, [body[i] for i in swaps]
    sw = swaps[0]
    assert body[sw].startswith("global_atomic_swap_x2") and body[sw].rstrip().endswith("sc0"), body[sw]   # returning form
    adds = [i for i, ins in enumerate(body) if re.match(r"global_atomic_add(_u32)? ", ins)]
    assert adds and all(body[i].rstrip().endswith("sc0") for i in adds), [body[i] for i in adds]
    ticket = min(i for i in adds if i > sw)
    between = body[sw + 1:ticket]
    assert "s_waitcnt vmcnt(0)" in between, between
    wait = sw + 1 + between.index("s_waitcnt vmcnt(0)")
    gap = body[sw + 1:wait]
    assert not any(i.startswith(("global_", "buffer_", "flat_", "scratch_", "s_cbranch", "s_branch")) for i in gap), gap
    assert not any(i < sw for i in adds), "a ticket add in front of the partial"
    # the last-ticket workgroup reads part[] with sc1 loads (never a plain load: that may be served by this CU's L1);
    # the first 64-bit load without sc1 behind them is DwpState::run, read by lane 0 after the sum
    first_store = min(i for i, ins in enumerate(body) if ins.startswith("global_store") and i > ticket)
    readers = [i for i, ins in enumerate(body) if ins.startswith("global_load_dwordx2") and ticket < i < first_store]
    assert readers and all(body[i].rstrip().endswith("sc1") for i in readers), [body[i] for i in readers]
    assert not any(ins.startswith("global_load_dwordx2") and ins.rstrip().endswith("sc1") for ins in body[:ticket])
