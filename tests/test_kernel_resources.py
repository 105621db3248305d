"""Register / scratch budget of the step's kernels at the benchmark shape, from the compiler's own resource report
(hipcc cross-compiles gfx950 without a GPU).  Two regressions of round 2 were invisible to the parity tests and cost
1.5 and 10 us per step: reduce_apply dropping from three to two 512-thread work-groups per CU (77 -> 89 VGPRs), and the plan
kernel's by-value arguments going through scratch memory after a run-time subscript into one of them."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fbtt-embedding_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]  # as __graft_entry__.build() compiles the product


def resources(src):
    """{mangled kernel name: {VGPRs, ScratchSize, Occupancy, ...}}"""
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-function", *PRELOAD,
                          "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull, os.path.join(CSRC, src)],
                         capture_output=True, text=True, timeout=900).stderr
    res, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return res


def pick(res, *needles):
    hits = [v for k, v in res.items() if all(n in k for n in needles)]
    assert len(hits) == 1, f"{needles}: {len(hits)} kernels match"
    return hits[0]


@pytest.fixture(scope="module")
def plan_res():
    return resources("ttx_plan.hip")


@pytest.fixture(scope="module")
def tt_res():
    # (the shape-specialised kernels are a translation unit per rank family)
    from concurrent.futures import ThreadPoolExecutor

    res = {}
    with ThreadPoolExecutor(4) as ex:  # (four compiler processes side by side: the fixture was 3.5 of the CPU suite's 6 minutes)
        for r in ex.map(resources, ("ttx_tt.hip", "ttx_tt_spec32.hip", "ttx_tt_spec64.hip", "ttx_tt_spec128a.hip")):
            res.update(r)
    return res


def test_plan_kernel_keeps_its_arguments_out_of_scratch(plan_res):
    for pro in ("ILb1E", "ILb0E"):
        r = pick(plan_res, "mb_single_kernel", pro)
        assert r["ScratchSize"] == 0, f"mb_single_kernel<{pro}>: {r}"
        assert r["VGPRs"] <= 128, r  # (a 1024-thread work-group is four waves per SIMD: 128 registers each)


def test_contraction_and_reduce_kernels_budget(tt_res):
    cfg2 = "Shape3ILi32ELi4ELi32ELi4ELi2ELi16ELi4EEELb0"  # (.., passes 2, chunk 16, q0 4), one sub-chunk
    # template flags after the shape: forward <MULTI, FUSE (pooling fused), PAD (padded shape)>, backward <MULTI, PAD>
    fwd, bwd = pick(tt_res, "spec_fwd_kernel", cfg2 + "ELb0ELb0E"), pick(tt_res, "spec_bwd_kernel", cfg2 + "ELb0E")
    fused = pick(tt_res, "spec_fwd_kernel", cfg2 + "ELb1ELb0E")  # ... the variant that pools the bags itself (ttx_tt_forward_o)
    padf, padb = pick(tt_res, "spec_fwd_kernel", cfg2 + "ELb0ELb1E"), pick(tt_res, "spec_bwd_kernel", cfg2 + "ELb1E")
    assert padf["ScratchSize"] == 0 and padb["ScratchSize"] == 0 and padb["VGPRs"] <= 128, (padf, padb)
    assert fwd["ScratchSize"] == 0 and bwd["ScratchSize"] == 0 and fused["ScratchSize"] == 0, (fwd, bwd, fused)
    assert fused["VGPRs"] <= 128, fused
    assert bwd["VGPRs"] <= 104, bwd  # four 256-thread work-groups per CU need <= 128; round 2 shipped 98, round 4 (MFMA tail) 76
    # the sub-chunked (large-batch) variants run with the NEXT sub-chunk's operands in flight: three work-groups per CU is what
    # their 46.8 KB of LDS allows, 168 registers is what that allows
    big = pick(tt_res, "spec_bwd_kernel", cfg2[:-1] + "1ELb0E")
    assert big["ScratchSize"] == 0 and big["VGPRs"] <= 168 and big["Occupancy"] >= 3, big
    bigf = pick(tt_res, "spec_fwd_kernel", cfg2[:-1] + "1ELb0ELb0E")
    assert bigf["ScratchSize"] == 0 and bigf["VGPRs"] <= 128, bigf
    d32 = "Shape3ILi32ELi4ELi32ELi4ELi2ELi32ELi2EEELb0"   # q0 = 2 (a select between a lane's two lookups once went to scratch)
    for k, d32k in (("spec_fwd_kernel", d32 + "ELb0ELb0E"), ("spec_bwd_kernel", d32 + "ELb0E")):
        r = pick(tt_res, k, d32k)
        assert r["ScratchSize"] == 0 and r["VGPRs"] <= 128, (k, r)
    # r = 128: one work-group per CU, the accumulator half of the register file in use -- and still nothing in scratch
    r128 = "Shape3ILi128ELi4ELi128ELi4ELi4ELi16ELi4EEELb0"
    for k, n in (("spec_fwd_kernel", r128 + "ELb0ELb0E"), ("spec_bwd_kernel", r128 + "ELb0E"), ("spec_bwd_kernel", r128 + "ELb1E")):
        r = pick(tt_res, k, n)
        assert r["ScratchSize"] == 0, (k, n, r)
    # r = 64 (cfg4; round 4): gradient rows staged per column pass leave 51.8 KB of LDS -- a THIRD work-group per CU, if the
    # backward holds 168 registers without spilling (the next core_1 block is fetched late in the pass for that)
    for r64 in ("Shape3ILi64ELi4ELi64ELi8ELi4ELi16ELi4EEELb0ELb0E", "Shape3ILi64ELi8ELi64ELi8ELi8ELi16ELi4EEELb0ELb0E",
                "Shape3ILi64ELi4ELi64ELi4ELi4ELi16ELi4EEELb0ELb0E"):
        r = pick(tt_res, "spec_bwd_kernel", r64)
        assert r["ScratchSize"] == 0 and r["VGPRs"] <= 168 and r["Occupancy"] >= 3, (r64, r)
    # round 5: q1 = 16 and q2 = 32 -- gradient rows per column pass with two / four float4 per lane (Shape3::NPL) keep two or three
    # work-groups on a CU; the q2 = 32 backward's 2 x 64 last-core registers stay out of scratch (exact AND padded variant: the
    # padded one is what a prime last factor runs on)
    for shp, occ in (("Shape3ILi32ELi16ELi32ELi16ELi8ELi16ELi4EEELb0ELb0E", 3), ("Shape3ILi32ELi16ELi32ELi16ELi8ELi16ELi4EEELb0ELb1E", 2),
                     ("Shape3ILi32ELi8ELi32ELi32ELi4ELi16ELi4EEELb0ELb0E", 2), ("Shape3ILi32ELi8ELi32ELi32ELi4ELi16ELi4EEELb0ELb1E", 2),
                     ("Shape3ILi32ELi4ELi32ELi32ELi2ELi16ELi4EEELb0ELb1E", 2), ("Shape3ILi32ELi16ELi32ELi32ELi8ELi16ELi4EEELb0ELb0E", 2)):
        r = pick(tt_res, "spec_bwd_kernel", shp)
        assert r["ScratchSize"] == 0 and r["Occupancy"] >= occ, (shp, r)
    # (the PADDED q = [4,16,32] backward -- q = [1,16,23] and friends -- spills 21 dwords at two work-groups per CU, like the padded
    #  r = 64 kernels: a second work-group is worth more than those)
    r = pick(tt_res, "spec_bwd_kernel", "Shape3ILi32ELi16ELi32ELi32ELi8ELi16ELi4EEELb0ELb1E")
    assert r["ScratchSize"] <= 128 and r["Occupancy"] >= 2, r
    # ... and the four-core gradient helper on the matrix pipe
    for q3 in ("ILi2ELi4E", "ILi4ELi4E", "ILi8ELi4E"):
        r = pick(tt_res, "t4_grad23_mfma_kernel", q3)
        assert r["ScratchSize"] == 0 and r["VGPRs"] <= 128, (q3, r)
    red = pick(tt_res, "reduce_apply_kernel")
    assert red["Occupancy"] >= 6, f"reduce_apply_kernel must leave room for three 512-thread work-groups per CU: {red}"
    assert pick(tt_res, "pool4_small_kernel")["ScratchSize"] == 0


@pytest.fixture(scope="module")
def spec32_asm():
    """gfx950 assembly listing of the rank-32 translation unit, compiled once for the disassembly assertions below"""
    out = os.path.join(ROOT, "build", "asm")
    os.makedirs(out, exist_ok=True)
    asm = os.path.join(out, "spec32_test.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-pass-failed", *PRELOAD,
                    "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "ttx_tt_spec32.hip")], check=True, timeout=900,
                   capture_output=True)
    return asm


def test_contraction_tails_stay_off_the_lds_pipe(spec32_asm):
    """Round 4: the forward tail is a reduce-scatter over the four lane quarters on v_permlane{32,16}_swap, both tails contract on
    v_mfma_f32_4x4x1, lookup records are fetched by the lane that needs them.  No ds_bpermute (LDS pipe, shared with the MFMA
    operand reads) and no v_cndmask butterfly may come back: rounds 1-3 spent 66 ds_bpermute + 128 v_cndmask per 64 MFMA there."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import asm_stats

    ks = asm_stats.parse_asm(spec32_asm)
    cfg2 = "Shape3ILi32ELi4ELi32ELi4ELi2ELi16ELi4EEE"
    seen = 0
    for name, c in ks.items():
        if cfg2 not in name or not c.get("total"):
            continue
        fwd = "spec_fwd_kernel" in name
        fused = fwd and "EEELb0ELb1E" in name  # (the fused-pooling variant hands bag counters round with three shuffles)
        if not fused:
            assert c.get("bpermute", 0) == 0, (name, c)
        padded = "ELb1EEEv" in name  # (the last template flag of both kernels)
        # a handful of address selects -- the padded variants pick an element's address or a valid dummy address per guarded
        # float4 --; the butterfly was 126.  (Round 6: + the selects that discard a record by its position in the sub-chunk: the
        # record stage fetches unconditionally so that nothing reads a load it has just issued.)
        assert c.get("cndmask", 0) <= (40 if padded else 24), (name, c)
        assert c.get("scratch", 0) == 0, (name, c)
        if fwd:
            assert c.get("permlane_swap", 0) >= 12, (name, c)
        assert c.get("mfma", 0) >= 96, (name, c)  # (64 + 32 small per group forward, 192 + 64 backward)
        seen += 1
    assert seen >= 6, seen


def _kernel_asm(asm_path, *needles):
    """the instruction lines of the one kernel of the listing whose mangled name holds every needle"""
    body, name, hits = [], None, {}
    for raw in open(asm_path):
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            name, body = m.group(1), []
            hits[name] = body
            continue
        if name is not None:
            ln = raw.strip()
            if ln.startswith("s_endpgm"):
                name = None
            elif ln and not ln.startswith((";", ".")):
                body.append(ln)
    sel = [v for k, v in hits.items() if all(n in k for n in needles)]
    assert len(sel) == 1, (needles, len(sel))
    return sel[0]


def test_sub_chunk_prologues_request_without_reading(spec32_asm):
    """Round 6: `s_waitcnt vmcnt` counts loads, stores and scratch in one in-order counter, and with stores pending the
    compiler waits for ZERO outstanding instructions at the first read of any loaded register -- a record fixed up behind its load, a
    default merged with it, a bag row fetched under `rec.x >= 0`, made the wave wait for every load it had just issued for the NEXT
    sub-chunk (three trips to memory in a row per sub-chunk of the backward, one of the forward).  In the large-batch kernels of the
    benchmark shape, between the first and the last global load of the prefetch block inside the sub-chunk loop there is no wait for
    vector memory on the path a plan with bag rows takes (the only wait sits in the branch of a plan without them, behind its
    dependent rowidx[] load, and in the per_sample_weights branch)."""
    asm = spec32_asm
    big = "Shape3ILi32ELi4ELi32ELi4ELi2ELi16ELi4EEELb1"
    for kern, flags, max_waits in (("spec_fwd_kernel", "ELb0ELb0E", 0), ("spec_bwd_kernel", "ELb0E", 2)):
        body = _kernel_asm(asm, kern, big + flags)
        # the prefetch block of the loop: the LAST run of global loads that is followed by a work-group barrier or the MFMAs of a
        # sub-chunk -- found as the loads between the kernel's last `s_barrier`-free stretch that starts with a dword load of a
        # record's row (ipos) ... simpler and robust: every maximal run of instructions between two MFMAs / barriers that holds
        # at least six global loads is a request block; none of them may hold more vmcnt waits than the optional branches explain
        runs, cur = [], []
        for ln in body:
            if ln.startswith(("v_mfma", "s_barrier")):
                if cur:
                    runs.append(cur)
                cur = []
            else:
                cur.append(ln)
        if cur:
            runs.append(cur)
        blocks = [r for r in runs if sum(x.startswith("global_load") for x in r) >= 6]
        assert blocks, kern
        assert len(blocks) >= 2, (kern, len(blocks))
        for r in blocks[1:]:  # (the first block is the kernel's entry: chunk record -> records -> operands is a dependent chain)
            first = next(i for i, x in enumerate(r) if x.startswith("global_load"))
            waits = [x for x in r[first:] if x.startswith("s_waitcnt") and "vmcnt" in x]  # (up to the block's end: the next MFMA / barrier)
            # (the kernels of round 5: one such wait in the forward, six in the backward)
            assert len(waits) <= max_waits, (kern, waits)


def test_contraction_kernels_get_their_first_pointers_with_the_wave(spec32_asm):
    """Round 6 (kernarg preload, ttx_tt_spec.inc TTX_KHEAD): the seven pointers the first three trips to memory need -- chunk
    records, lookup records, bag rows, plan header, the three cores -- lead the argument list as plain pointers, so the command
    processor hands them over in SGPRs (14 dwords, the most gfx950 preloads) and the chunk record is requested without waiting
    for an s_load of the argument segment."""
    text = open(spec32_asm).read()
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*spec_(?:fwd|bwd)_kernel\S*)(.*?)\.end_amdhsa_kernel", text, re.S):
        pl = re.search(r"\.amdhsa_user_sgpr_kernarg_preload_length (\d+)", m.group(2))
        assert pl and int(pl.group(1)) == 14, (m.group(1), pl and pl.group(1))
        seen += 1
    assert seen >= 4, seen


def _kernel_text(asm_path, *needles):
    """every instruction line of the one kernel whose mangled name holds every needle, early exits (s_endpgm) included"""
    text = open(asm_path).read()
    names = [n for n in re.findall(r"^(_Z\w+):", text, re.M) if all(x in n for x in needles)]
    assert len(names) == 1, (needles, len(names))
    beg = text.index("\n" + names[0] + ":")
    end = text.index(".Lfunc_end", beg)
    return [ln.strip() for ln in text[beg:end].splitlines()[2:] if ln.strip() and not ln.strip().startswith((";", "."))]


def test_single_sub_chunk_kernels_request_their_operands_in_one_round(spec32_asm):
    """Round 6: at the benchmark batch a work-group is one chain of trips to memory -- chunk record -> lookup records -> operands.
    The backward made five of them: the plan header word behind the chunk record, and two waits INSIDE the operand round (a
    component of a core-0 slice load copied into the register of its zero default; the gradient row times per_sample_weights
    meeting the plain row behind a branch).  Guard: between the records' request and the last operand request in front of the
    first MFMA there are at most two waits for vector memory on the path of a plan with bag rows and no weights (the records'
    own; the branch of a plan without bag rows holds a second), and the forward fetches its chunk record with ONE scalar load."""
    small = "Shape3ILi32ELi4ELi32ELi4ELi2ELi16ELi4EEELb0"
    for kern, flags in (("spec_bwd_kernel", "ELb0E"), ("spec_fwd_kernel", "ELb0ELb0E")):
        body = _kernel_text(spec32_asm, kern, small + flags)
        head = body[: next(i for i, x in enumerate(body) if x.startswith("v_mfma"))]
        rec = next(i for i, x in enumerate(head) if x.startswith("global_load_dwordx3"))  # the lookup records (int4 without .w)
        last = max(i for i, x in enumerate(head) if x.startswith("global_load"))
        assert sum(x.startswith("global_load") for x in head[rec:last + 1]) >= 8, kern
        waits = [x for x in head[rec:last] if x.startswith("s_waitcnt") and "vmcnt" in x]
        assert len(waits) <= 2, (kern, waits)
        if kern == "spec_fwd_kernel":
            chunk_loads = [x for x in head[:rec] if re.match(r"s_load_dword(x\d)? s\[?\d+(:\d+)?\]?, s\[2:3\]", x)]
            assert len(chunk_loads) == 1 and chunk_loads[0].startswith("s_load_dwordx4"), chunk_loads
