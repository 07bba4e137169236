#!/usr/bin/env python
"""Where does one wave's step go?  An in-order scoreboard model of a kernel's main loop, read from its ISA.

    bash scripts/tu_regs.sh l2o_unroll_pair.h 'k_unroll_pair<0, 1, 8, false, false>(UnrollPairArgs)' -- -mllvm -amdgpu-sched-strategy=max-ilp
    python scripts/isa_stall_model.py build/tu_regs.s _Z13k_unroll_pairILi0ELi1ELi8ELb0ELb0EEv14UnrollPairArgs

One wave per SIMD issues in order.  Every instruction of the step loop (the backward-branch region with the most MFMAs;
poll / timeout side branches fall through) gets an issue time = max(previous issue + the previous instruction's issue cost,
the ready time of its source registers, the free time of the pipe it needs) and a result-ready time = issue + latency.
The difference between its issue time and the earliest time the wave COULD have issued it (previous issue + cost) is a stall
of that instruction, charged to the class of the instruction that PRODUCED the late operand (or to the busy pipe).  That is
what SQ_WAIT_INST_ANY counts (a wave with an instruction that cannot issue); s_waitcnt / s_barrier time is SQ_WAIT_ANY and is
reported separately with nominal LDS latencies only (global memory and barriers are not modelled).

Costs (cycles; MI355X_MICROARCH.md "Per-instruction cycle constants", scripts/microbench/valu_issue_cost.hip,
two_wave_issue.hip): issue of a plain VALU instruction from a lone wave 5.26, transcendental 8.51, packed fp32 5.26, MFMA slot
5.26; bf16 16x16x32 MFMA occupies the matrix pipe 16.2 (back to back 17.9) and delivers its accumulator to a VALU reader
~40 after issue (8 passes x 4 + write-back), to a dependent MFMA on the same accumulator when the pipe is free; v_exp / v_rcp /
v_log / v_sqrt deliver 16 after issue; DPP / permlane consumers see VALU results 8 after issue; ds_read 64 (b32) .. 128 (b128)
after issue; SALU 4.  These are nominal -- the point is the ATTRIBUTION, which chains own the stall cycles, not the third digit."""
import re
import sys
from collections import defaultdict

ISSUE = {"valu": 5.26, "pk": 5.26, "trans": 8.51, "mfma": 5.26, "lds": 5.26, "salu": 4.0, "vmem": 5.26, "dpp": 5.26, "other": 4.0}
LAT = {"valu": 5.26, "pk": 5.26, "trans": 16.0, "mfma": 40.0, "lds": 110.0, "salu": 4.0, "vmem": 0.0, "dpp": 8.0, "other": 4.0}
# (vmem: not modelled -- the partner poll's round trip is s_waitcnt time, SQ_WAIT_ANY, and is measured by the phase clock)
MFMA_PIPE = 16.2
TRANS = ("v_exp_", "v_rcp_", "v_log_", "v_sqrt_", "v_rsq_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")) or "_dpp" in op:
        return "dpp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def regs(tok):
    """register names of one operand token: v12, v[4:7], s[2:3], a[0:3], vcc, exec"""
    out = []
    for kind, a, b in re.findall(r"\b([vsa])\[(\d+):(\d+)\]", tok):
        out += ["%s%d" % (kind, i) for i in range(int(a), int(b) + 1)]
    tok2 = re.sub(r"\b[vsa]\[\d+:\d+\]", "", tok)
    out += re.findall(r"\b[vsa]\d+\b", tok2)
    for special in ("vcc", "exec", "scc", "m0"):
        if re.search(r"\b%s\b" % special, tok2):
            out.append(special)
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    # SALU instructions EXECUTED per trip (PMC: SQ_INSTS_SALU / waves / steps): the loop region holds the poll / timeout side
    # branches too, which a step does not execute
    salu_executed = float(sys.argv[3]) if len(sys.argv) > 3 else None
    s = open(path).read()
    a = s.index(name + ":")
    b = s.index(".Lfunc_end", a)
    body = s[a:b].split("\n")
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lo = labels[m.group(1)]
            n = sum("v_mfma" in x for x in body[lo:i])
            if best is None or n > best[0] or (n == best[0] and i - lo < best[2] - best[1]):
                best = (n, lo, i)
    _, start, end = best
    insts = []
    for l in body[start:end]:
        t = l.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in rest.split(",")] if rest.strip() else []
        insts.append((op, ops, t))
    ready = defaultdict(float)          # register -> time its pending write lands
    producer = {}                       # register -> class of the instruction that wrote it last
    t_issue_prev, cost_prev = 0.0, 0.0
    mfma_free = 0.0
    lgkm = []                           # outstanding LDS / SMEM completions (in order)
    stall_by = defaultdict(float)
    stall_detail = defaultdict(float)
    issue_total = defaultdict(float)
    wait_total = 0.0
    n_by = defaultdict(int)
    phase, phase_stall = "head", defaultdict(float)
    mf = 0
    for op, ops, text in insts:
        cls = classify(op)
        n_by[cls] += 1
        earliest = t_issue_prev + cost_prev
        if op in ("s_barrier", "s_sleep", "s_endpgm") or op.startswith("s_cbranch") or op.startswith("s_branch"):
            t_issue_prev, cost_prev = earliest, ISSUE["salu"]
            continue
        if op == "s_waitcnt":
            t = earliest
            m = re.search(r"lgkmcnt\((\d+)\)", text)
            if m and lgkm:
                keep = int(m.group(1))
                need = lgkm[:len(lgkm) - keep] if keep < len(lgkm) else []
                if need:
                    t = max(t, max(need))
                lgkm = lgkm[len(lgkm) - keep:] if keep else []
            wait_total += t - earliest
            t_issue_prev, cost_prev = t, ISSUE["salu"]
            continue
        if op == "s_nop":
            k = int(ops[0]) + 1 if ops else 1
            issue_total["s_nop"] += 4.0 * k
            t_issue_prev, cost_prev = earliest, 4.0 * k
            continue
        # destination = first operand (stores / ds_write / cmp-to-vcc handled roughly: all operands are sources too)
        no_dest = op.startswith(("ds_write", "global_store", "global_atomic", "flat_store", "buffer_store", "s_cmp", "s_setprio",
                                 "s_setreg", "s_getreg"))
        dsts = [] if no_dest or not ops else regs(ops[0])
        srcs = []
        for k, o in enumerate(ops):
            if k == 0 and not no_dest and cls != "mfma" and not op.startswith(("v_fmac", "v_pk_fmac", "v_mac", "v_bfi", "v_cndmask")):
                continue
            srcs += regs(o)
        if op.startswith(("v_cmp", "v_cmpx")) and ops and not regs(ops[0]):
            dsts = ["vcc"]
        t = earliest
        why = None
        for r in srcs:
            rt = ready.get(r, 0.0)
            if producer.get(r) in ("valu", "pk") and cls == "dpp":
                rt = rt - LAT["valu"] + LAT["dpp"]
            if rt > t:
                t, why = rt, producer.get(r, "?")
        if cls == "mfma" and mfma_free > t:
            t, why = mfma_free, "matrix pipe busy"
        st = t - earliest
        if st > 0:
            key = "%s -> %s" % (why, cls)
            stall_by[key] += st
            phase_stall[phase] += st
        if cls == "mfma":
            mfma_free = t + MFMA_PIPE
            mf += 1
            phase = "MFMAs %d-%d" % (20 * ((mf - 1) // 20) + 1, 20 * ((mf - 1) // 20) + 20)
        issue_total[cls] += ISSUE[cls]
        lat = LAT[cls]
        if cls == "lds" and ("b32" in op):
            lat = 64.0
        for r in dsts:
            ready[r] = t + lat
            producer[r] = cls
        if cls == "lds" and not no_dest:
            lgkm.append(t + lat)
        if op.startswith("s_load"):
            lgkm.append(t + 200.0)
        t_issue_prev, cost_prev = t, ISSUE[cls]
    if salu_executed is not None and n_by["salu"]:
        issue_total["salu"] *= salu_executed / n_by["salu"]
    issue = sum(issue_total.values())
    total = issue + sum(stall_by.values()) + wait_total
    gap = sum((ISSUE[c] - 4.0) * n_by[c] for c in ("valu", "pk", "dpp", "mfma", "lds")) + (ISSUE["trans"] - 8.0) * n_by["trans"]
    stalls = sum(stall_by.values())
    print("%s\n  loop lines %d..%d: %d instructions  (%s)" % (name[:60], start, end, len(insts),
                                                          ", ".join("%s %d" % kv for kv in sorted(n_by.items(), key=lambda kv: -kv[1]))))
    print("  modelled straight-line time %.0f cycles = issue %.0f + issue stalls %.0f + s_waitcnt(lgkm, nominal LDS latency) %.0f"
          % (total, issue, stalls, wait_total))
    print("  of the issue time, %.0f cycles are the lone wave's issue gap (5.26 - 4 per VALU / MFMA / LDS instruction, 8.51 - 8 per "
          "transcendental): the SIMD has no second wave to issue from -- the hardware counts them as SQ_WAIT_INST_ANY, not as busy" % gap)
    print("  issue stalls by (producer -> stalled consumer):")
    for k, v in sorted(stall_by.items(), key=lambda kv: -kv[1]):
        if v >= 5:
            print("    %-34s %7.0f cycles" % (k, v))
    print("  issue stalls by position in the step (the MFMA block an instruction follows):")
    for k, v in phase_stall.items():
        print("    %-34s %7.0f cycles" % (k, v))


if __name__ == "__main__":
    main()
