"""Audit of compiled gfx950 code: for every matrix instruction, the distance in wait states to the first later
instruction that reads or writes one of its destination registers (straight-line code only: the scan of an instruction
ends at the next branch, barrier or end of program).

    hipcc ... -S --cuda-device-only kernels_attn.hip -o attn.s && python tools/mfma_hazard_scan.py attn.s

Reported per kernel:
  * VALU / LDS / memory instructions that touch a destination sooner than the matrix instruction's passes + 2 wait states
    (4-pass 16x16 forms: 6, 8-pass 32x32x16 fp16: 10, 16-pass 32x32x2 fp32: 18 - the distances the compiler's hazard
    recognizer pads to; `s_nop N` counts N + 1).  An inline-asm statement is not padded: any line printed here is a bug;
  * a count of matrix instructions that take an earlier one's whole destination as their C operand while writing other
    registers, 1-7 states later - the accumulator-forwarding case, which needs no padding and is listed for reference only.
Accumulators in AGPRs (a[..]) are not followed (build with -mllvm -amdgpu-mfma-vgpr-form=1, as the library is)."""
import re
import sys

PASSES = (("32x32x2", 16), ("32x32x16", 8), ("32x32x8", 8), ("16x16x32", 4), ("16x16x16", 4), ("16x16x4", 4), ("4x4", 2))


def regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        code = line.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        body.append(code)
        if code.startswith("s_endpgm"):
            yield name, body
            name = None


def scan(body):
    bugs, forwarded = [], 0
    for i, ins in enumerate(body):
        if not ins.startswith("v_mfma"):
            continue
        need = next((p for key, p in PASSES if key in ins), 16) + 2
        dst = regs(ins.split(None, 1)[1].split(",")[0])
        ws = 0
        for nxt in body[i + 1:i + 1 + need + 2]:
            if nxt.startswith("s_nop"):
                ws += int(nxt.split()[1]) + 1
                continue
            if nxt.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc")):
                break
            rest = nxt.split(None, 1)[1] if " " in nxt else ""
            if regs(rest) & dst:
                if nxt.startswith("v_mfma"):
                    ops = rest.split(",")
                    if regs(ops[3]) == dst and regs(ops[0]) != dst and not (regs(ops[1]) | regs(ops[2])) & dst:
                        forwarded += 1
                    elif (regs(ops[1]) | regs(ops[2])) & dst and ws < need:
                        bugs.append((ws, need, ins, nxt))
                elif ws < need:
                    bugs.append((ws, need, ins, nxt))
                break
            ws += 1
            if ws >= need:
                break
    return bugs, forwarded


def main():
    total = 0
    for path in sys.argv[1:]:
        for name, body in kernels(path):
            n_mfma = sum(ins.startswith("v_mfma") for ins in body)
            if not n_mfma:
                continue
            bugs, forwarded = scan(body)
            total += len(bugs)
            print(f"{name[:100]}: {n_mfma} matrix instructions, {forwarded} forwarded-C pairs, {len(bugs)} readers too close")
            for ws, need, ins, nxt in bugs[:8]:
                print(f"    {ws} of {need} wait states: {ins}   ->   {nxt}")
    print(f"total readers too close: {total}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
