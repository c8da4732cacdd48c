"""Tokeniser for the Rust subset (test infrastructure, see __init__.py)."""
import re

OPS = [
    "<<=", ">>=", "...", "..=", "::", "->", "=>", "==", "!=", "<=", ">=", "&&", "||",
    "+=", "-=", "*=", "/=", "%=", "^=", "&=", "|=", "<<", ">>", "..",
    "+", "-", "*", "/", "%", "^", "!", "&", "|", "=", "<", ">", "@", ".", ",", ";", ":",
    "#", "$", "?", "(", ")", "[", "]", "{", "}", "_",
]
_INT_SUFFIX = ("u8", "u16", "u32", "u64", "u128", "usize", "i8", "i16", "i32", "i64", "i128", "isize")
_ID = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")
_NUM = re.compile(r"0x[0-9a-fA-F_]+|0b[01_]+|0o[0-7_]+|[0-9][0-9_]*")
_FLOAT_TAIL = re.compile(r"\.[0-9][0-9_]*(?:[eE][+-]?[0-9_]+)?|[eE][+-]?[0-9_]+|\.(?![.A-Za-z_])")


class Tok:
    __slots__ = ("k", "v", "pos", "line")

    def __init__(self, k, v, pos, line):
        self.k, self.v, self.pos, self.line = k, v, pos, line

    def __repr__(self):
        return "%s(%r)@%d" % (self.k, self.v, self.line)


def lex(src):
    toks = []
    i, n, line = 0, len(src), 1
    while i < n:
        c = src[i]
        if c == "\n":
            line += 1
            i += 1
            continue
        if c in " \t\r":
            i += 1
            continue
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
            continue
        if src.startswith("/*", i):
            depth, j = 1, i + 2
            while depth and j < n:
                if src.startswith("/*", j):
                    depth += 1
                    j += 2
                elif src.startswith("*/", j):
                    depth -= 1
                    j += 2
                else:
                    if src[j] == "\n":
                        line += 1
                    j += 1
            i = j
            continue
        if c == '"' or (c == "b" and src.startswith('b"', i)) or (c == "r" and re.match(r'r#*"', src[i:])):
            if c == "r":
                m = re.match(r'r(#*)"', src[i:])
                end = '"' + m.group(1)
                j = src.index(end, i + len(m.group(0)))
                s = src[i + len(m.group(0)):j]
                j += len(end)
            else:
                j = i + (2 if c == "b" else 1)
                buf = []
                while src[j] != '"':
                    if src[j] == "\\":
                        buf.append(src[j:j + 2])
                        j += 2
                    else:
                        buf.append(src[j])
                        j += 1
                s = "".join(buf)
                j += 1
            line += src.count("\n", i, j)
            toks.append(Tok("str", s, i, line))
            i = j
            continue
        if c == "'":
            # char literal or lifetime
            m = re.match(r"'(\\.|\\x[0-9a-fA-F]{2}|\\u\{[0-9a-fA-F]+\}|[^'\\])'", src[i:])
            if m:
                toks.append(Tok("char", m.group(1), i, line))
                i += len(m.group(0))
                continue
            m = _ID.match(src, i + 1)
            toks.append(Tok("life", m.group(0), i, line))
            i = m.end()
            continue
        if c.isdigit():
            m = _NUM.match(src, i)
            txt = m.group(0)
            j = m.end()
            prev_dot = bool(toks) and toks[-1].k == "op" and toks[-1].v == "." and \
                not (len(toks) > 1 and toks[-2].k == "op" and toks[-2].v == ".")
            is_float = False
            if not prev_dot and not txt.startswith(("0x", "0b", "0o")):
                fm = _FLOAT_TAIL.match(src, j)
                if fm and not src.startswith("..", j):
                    # `1.method()` is not a float; `1.` / `1.5` / `1e3` are
                    is_float = True
                    txt += fm.group(0)
                    j = fm.end()
            sm = _ID.match(src, j)
            suffix = None
            if sm and sm.group(0) in _INT_SUFFIX + ("f32", "f64"):
                suffix = sm.group(0)
                j = sm.end()
            if is_float or suffix in ("f32", "f64"):
                toks.append(Tok("float", (float(txt.replace("_", "")), suffix), i, line))
            else:
                toks.append(Tok("int", (int(txt.replace("_", ""), 0) if txt[:2] in ("0x", "0b", "0o")
                                        else int(txt.replace("_", "")), suffix), i, line))
            i = j
            continue
        m = _ID.match(src, i)
        if m and not (m.group(0) == "_"):
            toks.append(Tok("id", m.group(0), i, line))
            i = m.end()
            continue
        for op in OPS:
            if src.startswith(op, i):
                toks.append(Tok("op", op, i, line))
                i += len(op)
                break
        else:
            raise SyntaxError("lex: unexpected %r at line %d" % (c, line))
    toks.append(Tok("eof", None, n, line))
    return toks
