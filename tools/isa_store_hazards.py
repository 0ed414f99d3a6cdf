"""Static check of a gfx950 .s file (hipcc -S --cuda-device-only): for every wide LDS write (ds_write_b96 / b128) and wide vector store
(buffer_ / global_ / scratch_store_dwordx3 / x4), how soon after it a VALU instruction writes one of its DATA registers.  The data of such
an instruction is read out over several cycles after issue; a VALU write inside that window ends up in the stored data (gemm256.hip, the
epilogue's HAZARD note; hipcc pads two wait states after wide buffer stores and none after wide LDS writes).  `s_nop N` counts as N + 1
states, everything else as one.
    python tools/isa_store_hazards.py file.s [min_states]   ->   one line per kernel with the closest distance found (and its count)"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


WIDE = re.compile(r"^(ds_write_b96|ds_write_b128|ds_write2_b64|(buffer|global|scratch)_store_dwordx[34])\b")
LOADS = re.compile(r"^(ds_read|ds_load|buffer_load|global_load|scratch_load|flat_load)")


def writes(ins):
    """VGPRs an instruction writes: the first operand of a VALU instruction (both operands of v_permlane*_swap and v_swap: they exchange
    their registers), the destination of an LDS / vector-memory LOAD whose data returns into VGPRs (an LDS-DMA load `... lds` has none)"""
    ops = [t.strip(",") for t in ins.split()[1:]]
    if not ops:
        return set()
    if ins.startswith("v_"):
        if ins.startswith("v_cmp") and not ins.startswith("v_cmpx"):
            return set()
        w = regs(ops[0])
        if re.match(r"v_(permlane\d+_swap|swap)", ins) and len(ops) > 1:
            w |= regs(ops[1])
        return w
    if LOADS.match(ins) and not ins.rstrip().endswith(" lds") and " lds" not in ins:
        return regs(ops[0])
    return set()


def scan(text, horizon=8):
    """{kernel: (closest distance in states or None, number of stores at that distance, example)}.  The walk behind a store follows
    fall-through code and, at an unconditional or conditional BACKWARD branch (a loop's back edge), continues at the branch target: the
    last stores of a loop body are checked against the first instructions of the next iteration."""
    out = {}
    for fn in re.findall(r"^(\w+):\s*;? ?@?\1", text, re.M) or re.findall(r"^([A-Za-z_]\w*):", text, re.M):
        i = text.index(fn + ":")
        j = text.find(".Lfunc_end", i)
        raw = [l.strip() for l in text[i:j].splitlines()]
        body, labels = [], {}
        for l in raw:
            m = re.match(r"^(\.L\w+):", l)
            if m:
                labels[m.group(1)] = len(body)              # index of the first instruction behind the label
                continue
            if l and not l.startswith(";") and not l.startswith("."):
                body.append(l)
        best, n, ex = None, 0, None
        for k, l in enumerate(body):
            if not WIDE.match(l):
                continue
            ops = [t.strip(",") for t in l.split()[1:]]
            data = regs(ops[1]) if l.startswith(("ds_write", "global_store")) else regs(ops[0])     # buffer_ / scratch_: data first
            if l.startswith("ds_write2") and len(ops) > 2:
                data |= regs(ops[2])
            dist, pos, steps = 0, k + 1, 0
            while pos < len(body) and steps < horizon:
                nx = body[pos]
                steps += 1
                dist += (int(nx.split()[1], 0) + 1) if nx.startswith("s_nop") else 1
                if nx.startswith(("s_endpgm", "s_barrier")):
                    break
                if nx.startswith(("s_cbranch", "s_branch")):
                    tgt = labels.get(nx.split()[1].strip(","))
                    if tgt is not None and tgt <= pos:       # back edge: the next iteration's head follows
                        pos = tgt
                        continue
                    if nx.startswith("s_branch"):
                        break                                # forward jump: not followed
                    pos += 1                                 # conditional forward branch: fall through
                    continue
                if writes(nx) & data:
                    if best is None or dist < best:
                        best, n, ex = dist, 1, l[:44] + "  ->  " + nx[:44]
                    elif dist == best:
                        n += 1
                    break
                pos += 1
        out[fn] = (best, n, ex)
    return out


def all_regs(ins):
    """every VGPR an instruction names (operands of any position)"""
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", ins):
        out |= regs(tok)
    return out


def scan_tr_reads(text):
    """The transposed LDS reads issued as inline assembly (gemm_tile.hpp tr_read_asm: `ds_read_b64_tr_b16` behind an asm statement):
    hipcc's wait-count pass does not know they are asynchronous, so the hand-placed `s_waitcnt lgkmcnt(0)` is what makes their
    destination registers valid.  -> {kernel: (number of such reads, [violations])}: a violation is an instruction that names a
    destination register of a transposed read before an `s_waitcnt` with lgkmcnt(0) has been passed (fall-through walk; a branch
    ends the walk as a violation unless the wait came first)."""
    out = {}
    for fn in re.findall(r"^(\w+):\s*;? ?@?\1", text, re.M) or re.findall(r"^([A-Za-z_]\w*):", text, re.M):
        i = text.index(fn + ":")
        j = text.find(".Lfunc_end", i)
        body, labels = [], {}
        for l in (x.strip() for x in text[i:j].splitlines()):
            m = re.match(r"^(\.L\w+):", l)
            if m:
                labels[m.group(1)] = len(body)
                continue
            if l and not l.startswith((";", ".")) and not re.match(r"^\w+:", l):
                body.append(l)

        def walk(pos, dst, hops):
            """first offending instruction on the paths from `pos`, or None: conditional branches are taken AND fallen through,
            unconditional ones followed (at most `hops` jumps per path)"""
            while pos < len(body):
                nx = body[pos]
                if nx.startswith("s_waitcnt") and re.search(r"lgkmcnt\(0\)", nx):
                    return None
                if nx.startswith("ds_read_b64_tr_b16"):
                    if regs(nx.split()[1].strip(",")) & dst:
                        return nx
                elif nx.startswith(("s_cbranch", "s_branch")):
                    tgt = labels.get(nx.split()[1].strip(","))
                    if tgt is None or hops == 0:
                        return nx
                    hit = walk(tgt, dst, hops - 1)
                    if hit is not None or nx.startswith("s_branch"):
                        return hit
                elif nx.startswith(("s_endpgm", "s_setpc")) or all_regs(nx) & dst:
                    return nx
                pos += 1
            return None
        n, bad = 0, []
        for k, l in enumerate(body):
            if l.startswith("ds_read_b64_tr_b16"):
                n += 1
                hit = walk(k + 1, regs(l.split()[1].strip(",")), 3)
                if hit is not None:
                    bad.append((l[:40], hit[:60]))
        out[fn] = (n, bad)
    return out


def self_test():
    """planted cases: every kind of writer the scan must see"""
    t = lambda body: scan_tr_reads("k:\n\t" + "\n\t".join(body) + "\n.Lfunc_end0:\n")["k"]
    assert t(["ds_read_b64_tr_b16 v[4:5], v1 offset:64", "v_mov_b32_e32 v9, v8", "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v9, v4"]) == (1, [])
    assert len(t(["ds_read_b64_tr_b16 v[4:5], v1", "v_mov_b32_e32 v9, v5", "s_waitcnt lgkmcnt(0)"])[1]) == 1          # read too early
    assert len(t(["ds_read_b64_tr_b16 v[4:5], v1", "s_waitcnt vmcnt(0)", "v_mfma_f32_16x16x32_bf16 a[0:3], v[4:7], v[8:11], a[0:3]"])[1]) == 1
    assert len(t(["ds_read_b64_tr_b16 v[4:5], v1", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v9, v4"])[1]) == 1          # a counted wait is not enough here
    assert t([".LBB0_1:", "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v9, v4", "ds_read_b64_tr_b16 v[4:5], v1", "s_cbranch_scc1 .LBB0_1",
              "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v9, v4"]) == (1, [])                                               # both ways out of a loop wait first
    assert len(t([".LBB0_1:", "v_mov_b32_e32 v9, v4", "ds_read_b64_tr_b16 v[4:5], v1", "s_cbranch_scc1 .LBB0_1", "s_waitcnt lgkmcnt(0)"])[1]) == 1
    k = lambda body: scan("k:\n\t" + "\n\t".join(body) + "\n.Lfunc_end0:\n")["k"][0]
    assert k(["ds_write_b128 v1, v[4:7]", "v_mov_b32_e32 v5, 0"]) == 1
    assert k(["buffer_store_dwordx4 v[4:7], v1, s[0:3], 0 offen", "s_nop 3", "v_add_f32_e32 v7, v1, v2"]) == 5
    assert k(["ds_write_b128 v1, v[4:7]", "v_permlane16_swap_b32_e32 v9, v6"]) == 1          # the swap writes BOTH operands
    assert k(["ds_write_b128 v1, v[4:7]", "ds_read_b128 v[6:9], v2"]) == 1                      # an LDS load's return
    assert k(["buffer_store_dwordx4 v[4:7], v1, s[0:3], 0 offen", "buffer_load_dwordx4 v[4:7], v2, s[0:3], 0 offen"]) == 1
    assert k(["ds_write_b128 v1, v[4:7]", "buffer_load_dwordx4 v3, s[0:3], 0 offen lds"]) is None   # LDS-DMA: no VGPR destination
    assert k([".LBB0_1:", "v_mov_b32_e32 v4, 0", "s_nop 0", "ds_write_b128 v1, v[4:7]", "s_cbranch_scc1 .LBB0_1"]) == 2   # over the back edge
    assert k(["ds_write_b128 v1, v[4:7]", "v_cmp_gt_f32_e32 vcc, v4, v5", "s_nop 7"]) is None
    return True


if __name__ == "__main__":
    self_test()
    res = scan(open(sys.argv[1]).read())
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for fn, (d, n, ex) in sorted(res.items(), key=lambda kv: (kv[1][0] is None, kv[1][0] or 0)):
        if d is not None and (not lim or d < lim):
            print("%-70s closest VALU overwrite of wide store data: %d states (%d times)  %s" % (fn[:70], d, n, ex))
    worst = min([d for d, _, _ in res.values() if d is not None], default=None)
    print("kernels: %d, closest distance: %s" % (len(res), worst))
