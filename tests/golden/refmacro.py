"""TEST INFRASTRUCTURE (used by tests/golden/gen_loop_decision_ref.py only).

The reference generates its tiled views (Tile / TileMut, TileBlocks*, TileRestoration*) with macro_rules!, which
tools/rustlite does not expand.  expand() does it textually: every macro of a file has ONE arm with `$x:ident`
parameters and one optional `$(,$opt:tt)?` tail; the expansion is the arm's body with the parameters substituted,
token for token, at each invocation.  unpointer() then rewrites the one idiom the transpiler cannot follow (see its
docstring).  Both operate on the text read from the reference tree at run time; nothing of it is stored here."""
import re


def _match_brace(t, i):
    """t[i] == '{' or '(' -> index just past its partner"""
    op = t[i]
    cl = {"{": "}", "(": ")", "[": "]"}[op]
    d = 0
    j = i
    while j < len(t):
        ch = t[j]
        if ch == "/" and t[j:j + 2] == "//":
            j = t.index("\n", j)
            continue
        if ch == op:
            d += 1
        elif ch == cl:
            d -= 1
            if d == 0:
                return j + 1
        j += 1
    raise ValueError("unbalanced")


def expand(text):
    macros = {}
    out = text
    while True:
        m = re.search(r"macro_rules!\s*([A-Za-z_0-9]+)\s*\{", out)
        if not m:
            break
        end = _match_brace(out, m.end() - 1)
        body = out[m.end():end - 1]
        # one arm: ( params ) => { body }
        pi = body.index("(")
        pe = _match_brace(body, pi)
        params = body[pi + 1:pe - 1]
        bi = body.index("{", body.index("=>", pe))
        be = _match_brace(body, bi)
        arm = body[bi + 1:be - 1]
        names = re.findall(r"\$([a-z_]+):ident", params)
        opt = re.search(r"\$\(\s*,\s*\$([a-z_]+):tt\s*\)\?", params)
        macros[m.group(1)] = (names, opt.group(1) if opt else None, arm)
        out = out[:m.start()] + out[end:]
    for name, (names, opt, arm) in macros.items():
        while True:
            m = re.search(r"\b%s!\s*\(" % name, out)
            if not m:
                break
            end = _match_brace(out, m.end() - 1)
            args = [a.strip() for a in out[m.end():end - 1].split(",") if a.strip()]
            body = arm
            optval = args[len(names)] if len(args) > len(names) else ""
            if opt:
                body = re.sub(r"\$\(\s*\$%s\s*\)\?" % opt, optval, body)
            for n, a in zip(names, args):
                body = re.sub(r"\$%s\b" % n, a, body)
            assert "$" not in body, body[body.index("$") - 40:body.index("$") + 40]
            e2 = end
            while e2 < len(out) and out[e2] in " \t":
                e2 += 1
            if e2 < len(out) and out[e2] == ";":
                e2 += 1
            out = out[:m.start()] + body + out[e2:]
    return out


def unpointer(text):
    """The tiled views keep `data` = a raw pointer to their first element and index rows by pointer arithmetic
    over the frame-level container (`self.data.add(index * stride)` + slice::from_raw_parts).  The transpiler has
    no pointer provenance for `&container[y][x]`, so the views here keep the frame-level container itself and
    index it with the absolute offsets they already carry (x, y) -- the same elements, no arithmetic changed."""
    t = text
    # constructors: the element reference -> the container
    t = re.sub(r"data:\s*&\s*(mut\s+)?frame_blocks\[y\]\[x\],", "data: frame_blocks,", t)
    t = re.sub(r"data:\s*&(mut\s+)?self\[y\]\[x\],", "data: self.data,", t)
    t = re.sub(r"data:\s*if x < frame_units\.cols && y < frame_units\.rows \{\s*&\s*(mut\s+)?frame_units\[y\]\[x\]\s*\} else \{[^}]*\},",
               "data: frame_units,", t)
    # row access
    t = re.sub(r"unsafe \{\s*let ptr = self\.data\.add\(index \* self\.(frame_cols|stride)\);\s*slice::from_raw_parts(_mut)?\(ptr, self\.cols\)\s*\}",
               lambda m: "&%sself.data[self.y + index][self.x..self.x + self.cols]" % ("mut " if m.group(2) else ""), t)
    assert "from_raw_parts" not in t and ".add(" not in t, "pointer idiom left"
    return t
