"""Build step: no packed-fp32 instruction of libsp3d.so takes its LOW result from the HIGH half of source 1.

Why (tools/mfma_pk_hazard5.hip, profiles/r05_mfma_pk_hazard5.txt): on MI355X a `v_pk_{fma,mul,add}_f32` whose op_sel bit of
source 1 is set (low result <- high half of src1) returns, in lanes 48-63, the low result computed with that operand read as
ZERO whenever waves of ANOTHER kernel on the same CU are executing the double-rate matrix instructions
(v_mfma_f32_16x16x32_{bf16,f16}, v_mfma_f32_32x32x16_bf16, v_mfma_i32_16x16x64_i8) - two streams or two processes on one GPU.
The same selection on source 0 or source 2, and high result <- low half on any source, are not affected; nor are neighbours
running fp32 / f64 / the older 16x16x16 matrix instructions.  The compiler picks the operand order and the op_sel bits of the
packed instructions it forms from 2-wide vector arithmetic (sp3d_proj_pk.h), so the build compiles every source to assembly,
rewrites the affected instructions here - multiplication and addition commute, so source 0 and source 1 trade places together
with their op_sel / op_sel_hi / neg_lo / neg_hi bits - and assembles the result.  Same operations, same bits.

`count_risky` is also what tests/test_host_cabi.py runs over the disassembly of the finished library.
"""
from __future__ import annotations

import re

_PK = re.compile(r"^(\s*)(v_pk_(?:fma|mul|add)_f32)(?:_e64)?\s+(.*?)\s*(//.*|;.*)?$")
_MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,\s]+)\]")


def _split_operands(s: str):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _parse(line: str):
    """-> (indent, mnemonic, operands, {modifier: bits}, other words, comment) of a packed-fp32 instruction, else None"""
    m = _PK.match(line)
    if not m:
        return None
    indent, mnem, rest, comment = m.group(1), m.group(2), m.group(3), m.group(4) or ""
    mods = {k: [int(b) for b in v.replace(" ", "").split(",")] for k, v in _MOD.findall(rest)}
    rest = _MOD.sub("", rest)
    nsrc = 3 if mnem == "v_pk_fma_f32" else 2
    ops = _split_operands(rest)
    words = ""
    if len(ops) == nsrc + 1:                                   # "dst, s0, s1[, s2] clamp" : trailing words ride on the last operand
        last = ops[nsrc].split(None, 1)
        ops[nsrc] = last[0]
        words = last[1] if len(last) > 1 else ""
    if len(ops) != nsrc + 1:
        raise ValueError(f"cannot parse packed instruction: {line!r}")
    return indent, mnem, ops, mods, words.strip(), comment


def _risky(parsed) -> bool:
    op_sel = parsed[3].get("op_sel")
    return bool(op_sel and len(op_sel) > 1 and op_sel[1] == 1)


def count_risky(asm_text: str) -> int:
    """packed-fp32 instructions whose low result reads the high half of source 1 (assembly or llvm-objdump text)"""
    n = 0
    for line in asm_text.splitlines():
        if "v_pk_" not in line:
            continue
        body = line.split("//")[0]
        body = re.sub(r"^\s*[0-9a-fA-F]+:\s*", "\t", body) if re.match(r"^\s*[0-9a-fA-F]+:\s", body) else body
        p = _parse(body)
        if p and _risky(p):
            n += 1
    return n


def fix_asm(asm_text: str):
    """-> (assembly with source 0 / source 1 exchanged in every affected instruction, number rewritten)"""
    out, n = [], 0
    for line in asm_text.splitlines():
        p = _parse(line) if "v_pk_" in line else None
        if p is None or not _risky(p):
            out.append(line)
            continue
        indent, mnem, ops, mods, words, comment = p
        nsrc = len(ops) - 1
        full = {"op_sel": [0] * nsrc, "op_sel_hi": [1] * nsrc, "neg_lo": [0] * nsrc, "neg_hi": [0] * nsrc}
        for k, v in mods.items():
            if len(v) != nsrc:
                raise ValueError(f"modifier {k} of {line!r} does not have {nsrc} entries")
            full[k] = list(v)
        if full["op_sel"][0] == 1:
            raise ValueError(f"both multiplicands take their low result from a high half, exchanging them does not help: {line!r}")
        ops[1], ops[2] = ops[2], ops[1]
        for v in full.values():
            v[0], v[1] = v[1], v[0]
        text = f"{indent}{mnem} " + ", ".join(ops)
        for k, default in (("op_sel", 0), ("op_sel_hi", 1), ("neg_lo", 0), ("neg_hi", 0)):
            if any(b != default for b in full[k]):
                text += f" {k}:[" + ",".join(str(b) for b in full[k]) + "]"
        if words:
            text += " " + words
        if comment:
            text += " " + comment
        out.append(text)
        n += 1
    return "\n".join(out) + "\n", n
