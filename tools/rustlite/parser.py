"""Parser for the Rust subset (test infrastructure, see __init__.py).

Items are parsed eagerly (signatures, struct fields, enum variants, consts);
function bodies are kept as token ranges and parsed on first use, so syntax
that only occurs in functions nobody asks for never has to be supported.
"""
from .lexer import lex, Tok


class N:
    """AST node: kind + attributes."""

    def __init__(self, k, **kw):
        self.k = k
        self.__dict__.update(kw)

    def __repr__(self):
        return "N(%s)" % ", ".join("%s=%r" % kv for kv in self.__dict__.items())


class ParseError(SyntaxError):
    pass


BLOCK_LIKE = ("if", "match", "loop", "while", "for", "block", "unsafe")
ASSIGN_OPS = ("=", "+=", "-=", "*=", "/=", "%=", "^=", "&=", "|=", "<<=", ">>=")
BINPREC = {
    "*": 11, "/": 11, "%": 11, "+": 10, "-": 10, "<<": 9, ">>": 9, "&": 8, "^": 7, "|": 6,
    "==": 5, "!=": 5, "<": 5, ">": 5, "<=": 5, ">=": 5, "&&": 4, "||": 3,
}


class Parser:
    def __init__(self, toks, fname="?"):
        self.t = toks
        self.i = 0
        self.fname = fname

    # ---- token helpers ----
    def peek(self, o=0):
        return self.t[min(self.i + o, len(self.t) - 1)]

    def err(self, msg):
        tk = self.peek()
        raise ParseError("%s:%d: %s (at %r)" % (self.fname, tk.line, msg, tk.v))

    def at(self, v, o=0):
        tk = self.peek(o)
        return tk.k in ("op", "id") and tk.v == v

    def at_id(self, o=0):
        return self.peek(o).k == "id"

    def eat(self, v):
        if self.at(v):
            self.i += 1
            return True
        return False

    def expect(self, v):
        if not self.eat(v):
            self.err("expected %r" % v)

    def ident(self):
        tk = self.peek()
        if tk.k != "id":
            self.err("expected identifier")
        self.i += 1
        return tk.v

    def split_shift(self):
        """`>>` / `>=` / `>>=` seen where one `>` closes generics."""
        tk = self.peek()
        if tk.k == "op" and tk.v in (">>", ">=", ">>="):
            rest = tk.v[1:]
            self.t[self.i] = Tok("op", ">", tk.pos, tk.line)
            self.t.insert(self.i + 1, Tok("op", rest, tk.pos + 1, tk.line))

    def skip_attrs(self):
        attrs = []
        while self.at("#"):
            self.i += 1
            self.eat("!")
            start = self.i
            self.skip_balanced("[", "]")
            attrs.append(" ".join(str(t.v) for t in self.t[start:self.i]))
        return attrs

    def skip_balanced(self, o, c):
        if not self.at(o):
            self.err("expected %r" % o)
        depth = 0
        while True:
            tk = self.peek()
            if tk.k == "eof":
                self.err("unbalanced %s" % o)
            if tk.k == "op" and tk.v == o:
                depth += 1
            elif tk.k == "op" and tk.v == c:
                depth -= 1
                if depth == 0:
                    self.i += 1
                    return
            self.i += 1

    # ---- types ----
    def parse_type(self):
        if self.eat("&") or self.eat("&&"):
            if self.peek().k == "life":
                self.i += 1
            mut = self.eat("mut")
            return N("tref", mut=mut, inner=self.parse_type())
        if self.at("*") and (self.at("const", 1) or self.at("mut", 1)):
            self.i += 2
            return N("tptr", inner=self.parse_type())
        if self.eat("["):
            el = self.parse_type()
            if self.eat(";"):
                n = self.parse_expr()
                self.expect("]")
                return N("tarray", el=el, n=n)
            self.expect("]")
            return N("tslice", el=el)
        if self.eat("("):
            els = []
            while not self.at(")"):
                els.append(self.parse_type())
                if not self.eat(","):
                    break
            self.expect(")")
            return N("ttuple", els=els)
        if self.eat("!"):
            return N("tnever")
        if self.at("impl") or self.at("dyn"):
            self.i += 1
            self.parse_bounds()
            return N("topaque")
        if self.at("unsafe") or self.at("extern") or self.at("fn"):
            self.eat("unsafe")
            if self.eat("extern"):
                if self.peek().k == "str":
                    self.i += 1
            self.expect("fn")
            self.skip_balanced("(", ")")
            if self.eat("->"):
                self.parse_type()
            return N("tfn")
        if self.eat("_"):
            return N("tinfer")
        if self.at("<"):  # <T as Trait>::Assoc
            self.i += 1
            t = self.parse_type()
            if self.eat("as"):
                self.parse_type()
            self.split_shift()
            self.expect(">")
            segs = [t]
            while self.eat("::"):
                segs.append(self.ident())
            return N("tqpath", base=t, rest=segs[1:])
        return self.parse_path(in_type=True)

    def parse_bounds(self):
        """-> the trait names of the bound list (last path segment each)"""
        names = []
        while True:
            self.eat("?")
            if self.peek().k == "life":
                self.i += 1
            elif self.at("("):
                self.skip_balanced("(", ")")
            else:
                pth = self.parse_path(in_type=True)
                if getattr(pth, "segs", None):
                    names.append(pth.segs[-1])
                if self.at("("):  # Fn(A) -> B
                    self.skip_balanced("(", ")")
                    if self.eat("->"):
                        self.parse_type()
            if not self.eat("+"):
                break
        return names

    def parse_generic_args(self):
        """after `<`; returns list of N (types / const exprs / lifetimes dropped)."""
        args = []
        while True:
            self.split_shift()
            if self.at(">"):
                break
            tk = self.peek()
            if tk.k == "life":
                self.i += 1
            elif tk.k in ("int",) or self.at("-"):
                args.append(N("gconst", e=self.parse_unary()))
            elif self.at("{"):
                self.i += 1
                e = self.parse_expr()
                self.expect("}")
                args.append(N("gconst", e=e))
            elif tk.k == "id" and self.at("=", 1):  # assoc binding
                self.i += 2
                self.parse_type()
            else:
                args.append(self.parse_type())
            if not self.eat(","):
                break
        self.split_shift()
        self.expect(">")
        return args

    def parse_path(self, in_type=False):
        """a::b::<T>::c ; in types `a::b<T>`."""
        segs = []
        gen = {}
        self.eat("::")
        while True:
            tk = self.peek()
            if tk.k != "id":
                self.err("expected path segment")
            self.i += 1
            segs.append(tk.v)
            if in_type and self.at("<"):
                self.i += 1
                gen[len(segs) - 1] = self.parse_generic_args()
            elif in_type and self.at("(") and segs[-1] in ("Fn", "FnMut", "FnOnce"):
                self.skip_balanced("(", ")")
                if self.eat("->"):
                    self.parse_type()
            if self.at("::") and self.at("<", 1):
                self.i += 2
                gen[len(segs) - 1] = self.parse_generic_args()
            if self.at("::") and self.peek(1).k == "id":
                self.i += 1
                continue
            break
        return N("path", segs=segs, gen=gen)

    def parse_generics_decl(self):
        """`<T: Pixel, const N: usize, 'a>` -> list of (kind, name)."""
        out = []
        if not self.eat("<"):
            return out
        while True:
            self.split_shift()
            if self.at(">"):
                break
            self.skip_attrs()
            tk = self.peek()
            if tk.k == "life":
                self.i += 1
                if self.eat(":"):
                    while self.peek().k == "life":
                        self.i += 1
                        if not self.eat("+"):
                            break
            elif self.eat("const"):
                name = self.ident()
                self.expect(":")
                ty = self.parse_type()
                if self.eat("="):
                    self.parse_unary()
                out.append(("const", name, ty))
            else:
                name = self.ident()
                bounds = self.parse_bounds() if self.eat(":") else []
                if self.eat("="):
                    self.parse_type()
                out.append(("type", name, tuple(bounds)))
            if not self.eat(","):
                break
        self.split_shift()
        self.expect(">")
        return out

    def skip_where(self):
        if self.eat("where"):
            while not (self.at("{") or self.at(";")):
                self.i += 1

    # ---- items ----
    def parse_items(self, until=None):
        items = []
        while True:
            if until and self.at(until):
                break
            if self.peek().k == "eof":
                break
            it = self.parse_item()
            if it is not None:
                items.append(it)
        return items

    def parse_vis(self):
        if self.eat("pub"):
            if self.at("("):
                self.skip_balanced("(", ")")

    def parse_item(self):
        attrs = self.skip_attrs()
        self.parse_vis()
        tk = self.peek()
        if tk.k != "id":
            if self.eat(";"):
                return None
            self.err("expected item")
        v = tk.v
        if v == "use":
            while not self.eat(";"):
                self.i += 1
            return None
        if v == "extern" and self.peek(1).k == "str" and self.at("{", 2):
            self.i += 2
            self.skip_balanced("{", "}")
            return None
        if v == "extern" and self.at("{", 1):
            self.i += 1
            self.skip_balanced("{", "}")
            return None
        if v == "extern" and self.at("crate", 1):
            while not self.eat(";"):
                self.i += 1
            return None
        if v == "mod":
            self.i += 1
            name = self.ident()
            if self.eat(";"):
                return None
            self.expect("{")
            items = self.parse_items(until="}")
            self.expect("}")
            return N("mod", name=name, items=items, attrs=attrs)
        if v in ("fn", "const", "unsafe", "async", "extern") and self._is_fn_ahead():
            return self.parse_fn(attrs)
        if v in ("const", "static"):
            self.i += 1
            self.eat("mut")
            name = self.ident() if not self.eat("_") else "_"
            self.expect(":")
            ty = self.parse_type()
            init = None
            if self.eat("="):
                init = self.parse_expr()
            self.expect(";")
            return N("const", name=name, ty=ty, init=init, attrs=attrs)
        if v == "struct" or v == "union":
            self.i += 1
            name = self.ident()
            self.parse_generics_decl()
            fields = []
            tuple_like = False
            self.skip_where()
            if self.eat("{"):
                while not self.at("}"):
                    self.skip_attrs()
                    self.parse_vis()
                    fn_ = self.ident()
                    self.expect(":")
                    fields.append((fn_, self.parse_type()))
                    if not self.eat(","):
                        break
                self.expect("}")
            elif self.eat("("):
                tuple_like = True
                k = 0
                while not self.at(")"):
                    self.skip_attrs()
                    self.parse_vis()
                    fields.append((str(k), self.parse_type()))
                    k += 1
                    if not self.eat(","):
                        break
                self.expect(")")
                self.skip_where()
                self.expect(";")
            else:
                self.expect(";")
            return N("struct", name=name, fields=fields, tuple_like=tuple_like, attrs=attrs)
        if v == "enum":
            self.i += 1
            name = self.ident()
            self.parse_generics_decl()
            self.skip_where()
            self.expect("{")
            variants = []
            nextd = 0
            default_variant = None
            while not self.at("}"):
                vattrs = self.skip_attrs()
                vn = self.ident()
                if any("default" in a.split() for a in vattrs):
                    default_variant = vn       # #[default] of #[derive(Default)]
                kind, fields = "unit", []
                if self.at("("):
                    kind = "tuple"
                    self.i += 1
                    while not self.at(")"):
                        fields.append(self.parse_type())
                        if not self.eat(","):
                            break
                    self.expect(")")
                elif self.at("{"):
                    kind = "struct"
                    self.i += 1
                    while not self.at("}"):
                        self.skip_attrs()
                        f = self.ident()
                        self.expect(":")
                        fields.append((f, self.parse_type()))
                        if not self.eat(","):
                            break
                    self.expect("}")
                disc = None
                if self.eat("="):
                    disc = self.parse_expr()
                variants.append((vn, kind, fields, disc))
                if not self.eat(","):
                    break
            self.expect("}")
            return N("enum", name=name, variants=variants, attrs=attrs, default_variant=default_variant)
        if v == "impl" or (v == "unsafe" and self.at("impl", 1)):
            self.eat("unsafe")
            self.i += 1
            gens = self.parse_generics_decl()
            self.eat("!")
            t1 = self.parse_type()
            trait = None
            if self.eat("for"):
                trait = t1
                t1 = self.parse_type()
            self.skip_where()
            self.expect("{")
            items = self.parse_items(until="}")
            self.expect("}")
            return N("impl", ty=t1, trait=trait, items=items, gens=gens, attrs=attrs)
        if v == "trait" or (v == "unsafe" and self.at("trait", 1)):
            self.eat("unsafe")
            self.i += 1
            name = self.ident()
            gens = self.parse_generics_decl()
            if self.eat(":"):
                self.parse_bounds()
            self.skip_where()
            self.expect("{")
            items = self.parse_items(until="}")
            self.expect("}")
            return N("trait", name=name, items=items, gens=gens, attrs=attrs)
        if v == "type":
            self.i += 1
            name = self.ident()
            self.parse_generics_decl()
            if self.eat(":"):
                self.parse_bounds()
            ty = None
            if self.eat("="):
                ty = self.parse_type()
            self.skip_where()
            self.expect(";")
            return N("alias", name=name, ty=ty, attrs=attrs)
        if v == "macro_rules":
            self.i += 2
            name = self.ident()
            start = self.i
            if self.at("{"):
                self.skip_balanced("{", "}")
            else:
                self.skip_balanced("(", ")")
                self.eat(";")
            return N("macro_rules", name=name, toks=(start, self.i), attrs=attrs)
        o = 0
        while self.at("::", o + 1) and self.peek(o + 2).k == "id":
            o += 2
        if self.at("!", o + 1):  # item-position macro call (possibly path-qualified)
            self.i += o
            name = self.ident()
            self.i += 1
            start = self.i
            if self.at("{"):
                self.skip_balanced("{", "}")
            elif self.at("("):
                self.skip_balanced("(", ")")
                self.eat(";")
            else:
                self.skip_balanced("[", "]")
                self.eat(";")
            return N("item_macro", name=name, toks=(start, self.i), attrs=attrs)
        self.err("unsupported item")

    def _is_fn_ahead(self):
        o = 0
        while self.at("const", o) or self.at("unsafe", o) or self.at("async", o) or self.at("extern", o):
            o += 1
            if self.peek(o).k == "str":
                o += 1
        return self.at("fn", o)

    def parse_fn(self, attrs):
        while not self.at("fn"):
            self.i += 1
        self.i += 1
        name = self.ident()
        gens = self.parse_generics_decl()
        self.expect("(")
        params = []
        self_kind = None
        while not self.at(")"):
            self.skip_attrs()
            # self forms
            save = self.i
            if self.at("self") or (self.at("mut") and self.at("self", 1)) or \
               (self.at("&") and (self.at("self", 1) or (self.at("mut", 1) and self.at("self", 2))
                                  or (self.peek(1).k == "life" and (self.at("self", 2) or self.at("self", 3))))):
                while not self.at("self"):
                    self.i += 1
                self.i += 1
                self_kind = "self"
                if self.eat(":"):
                    self.parse_type()
            else:
                self.i = save
                pat = self.parse_pattern()
                self.expect(":")
                ty = self.parse_type()
                params.append((pat, ty))
            if not self.eat(","):
                break
        self.expect(")")
        ret = None
        if self.eat("->"):
            ret = self.parse_type()
        self.skip_where()
        body = None
        if self.at("{"):
            start = self.i
            self.skip_balanced("{", "}")
            body = (start, self.i)
        else:
            self.expect(";")
        return N("fn", name=name, gens=gens, params=params, ret=ret, body=body, has_self=self_kind is not None,
                 attrs=attrs, parsed=None)

    def parse_body(self, rng):
        """Parse a function body given its token range (lazily)."""
        sub = Parser(self.t[rng[0]:rng[1]] + [Tok("eof", None, 0, self.t[rng[1] - 1].line)], self.fname)
        blk = sub.parse_block()
        blk.parser = sub
        return blk

    # ---- patterns ----
    def parse_pattern(self):
        self.eat("|")
        alts = [self.parse_pattern1()]
        while self.at("|") and not self.at("||"):
            self.i += 1
            alts.append(self.parse_pattern1())
        return alts[0] if len(alts) == 1 else N("por", alts=alts)

    def parse_pattern1(self):
        tk = self.peek()
        if self.eat("_"):
            return N("pwild")
        if self.eat("&") or self.eat("&&"):
            self.eat("mut")
            return N("pref", inner=self.parse_pattern1())
        if self.eat("("):
            els = []
            while not self.at(")"):
                els.append(self.parse_pattern())
                if not self.eat(","):
                    break
            self.expect(")")
            if len(els) == 1 and self.t[self.i - 2].v != ",":
                return els[0]
            return N("ptuple", els=els)
        if self.eat("["):
            els = []
            while not self.at("]"):
                if self.eat(".."):
                    els.append(N("prest"))
                else:
                    els.append(self.parse_pattern())
                if not self.eat(","):
                    break
            self.expect("]")
            return N("pslice", els=els)
        if tk.k in ("int", "float", "str", "char") or self.at("-"):
            lo = self.parse_lit_pat()
            if self.at("..=") or self.at("...") or self.at(".."):
                incl = not self.at("..")
                self.i += 1
                hi = None
                if self.peek().k in ("int", "char") or self.at("-") or self.at_id():
                    hi = self.parse_lit_pat()
                return N("prange", lo=lo, hi=hi, incl=incl)
            return N("plit", e=lo)
        if self.eat(".."):
            return N("prest")
        if tk.k == "id":
            if tk.v in ("ref", "mut"):
                byref = self.eat("ref")
                mut = self.eat("mut")
                name = self.ident()
                sub = None
                if self.eat("@"):
                    sub = self.parse_pattern1()
                return N("pbind", name=name, sub=sub, mut=mut)
            if tk.v in ("true", "false"):
                self.i += 1
                return N("plit", e=N("bool", v=tk.v == "true"))
            path = self.parse_path()
            if self.at("("):
                self.i += 1
                els = []
                while not self.at(")"):
                    if self.eat(".."):
                        els.append(N("prest"))
                    else:
                        els.append(self.parse_pattern())
                    if not self.eat(","):
                        break
                self.expect(")")
                return N("ptstruct", path=path, els=els)
            if self.at("{"):
                self.i += 1
                fields = []
                rest = False
                while not self.at("}"):
                    if self.eat(".."):
                        rest = True
                        break
                    self.eat("ref")
                    self.eat("mut")
                    f = self.ident() if self.peek().k == "id" else str(self.peek().v[0])
                    if self.eat(":"):
                        fields.append((f, self.parse_pattern()))
                    else:
                        fields.append((f, N("pbind", name=f, sub=None)))
                    if not self.eat(","):
                        break
                self.expect("}")
                return N("pstruct", path=path, fields=fields)
            if len(path.segs) == 1 and not path.gen:
                name = path.segs[0]
                if self.eat("@"):
                    return N("pbind", name=name, sub=self.parse_pattern1())
                if self.at("..=") or self.at("..."):
                    self.i += 1
                    hi = self.parse_lit_pat()
                    return N("prange", lo=N("pathx", path=path), hi=hi, incl=True)
                # could be a const / unit variant: decided by the transpiler
                return N("pident", name=name)
            if self.at("..=") or self.at("..."):
                self.i += 1
                hi = self.parse_lit_pat()
                return N("prange", lo=N("pathx", path=path), hi=hi, incl=True)
            return N("ppath", path=path)
        self.err("unsupported pattern")

    def parse_lit_pat(self):
        neg = self.eat("-")
        tk = self.peek()
        if tk.k == "int":
            self.i += 1
            e = N("int", v=tk.v[0], suffix=tk.v[1])
        elif tk.k == "float":
            self.i += 1
            e = N("float", v=tk.v[0], suffix=tk.v[1])
        elif tk.k == "str":
            self.i += 1
            e = N("str", v=tk.v)
        elif tk.k == "char":
            self.i += 1
            e = N("char", v=tk.v)
        elif tk.k == "id":
            e = N("pathx", path=self.parse_path())
        else:
            self.err("bad literal pattern")
        return N("unary", op="-", e=e) if neg else e

    # ---- blocks / statements ----
    def parse_block(self):
        self.expect("{")
        stmts = []
        tail = None
        while not self.at("}"):
            attrs = self.skip_attrs()
            if self.eat(";"):
                continue
            if self.at("let"):
                self.i += 1
                pat = self.parse_pattern()
                ty = None
                if self.eat(":"):
                    ty = self.parse_type()
                init = None
                els = None
                if self.eat("="):
                    init = self.parse_expr()
                    if self.eat("else"):
                        els = self.parse_block()
                self.expect(";")
                stmts.append(N("let", pat=pat, ty=ty, init=init, els=els, line=self.peek().line))
                continue
            if self._at_item():
                it = self.parse_item()
                if it is not None:
                    stmts.append(N("item", item=it))
                continue
            line = self.peek().line
            e = self.parse_expr(stmt=True)
            if self.eat(";"):
                stmts.append(N("expr", e=e, line=line))
            elif self.at("}"):
                tail = e
            elif e.k in BLOCK_LIKE or (e.k == "macro" and e.brace):
                stmts.append(N("expr", e=e, line=line))
            else:
                self.err("expected ; or }")
        self.expect("}")
        return N("block", stmts=stmts, tail=tail)

    def _at_item(self):
        tk = self.peek()
        if tk.k != "id":
            return False
        if tk.v in ("fn", "struct", "enum", "impl", "trait", "use", "mod", "static", "type", "macro_rules", "pub"):
            return True
        if tk.v == "const" and not self.at("{", 1):
            return True
        if tk.v in ("unsafe", "extern") and self._is_fn_ahead():
            return True
        return False

    # ---- expressions ----
    def parse_expr(self, no_struct=False, stmt=False):
        return self.parse_assign(no_struct, stmt)

    def parse_assign(self, ns, stmt=False):
        lhs = self.parse_range(ns, stmt)
        tk = self.peek()
        if tk.k == "op" and tk.v in ASSIGN_OPS:
            self.i += 1
            rhs = self.parse_assign(ns)
            return N("assign", op=tk.v, l=lhs, r=rhs)
        return lhs

    def _expr_start(self):
        tk = self.peek()
        if tk.k in ("int", "float", "str", "char", "id", "life"):
            return not (tk.k == "id" and tk.v in ("as", "else", "in"))
        return tk.k == "op" and tk.v in ("(", "[", "{", "-", "!", "*", "&", "&&", "|", "||", "<", "::")

    def parse_range(self, ns, stmt=False):
        if self.at("..") or self.at("..="):
            incl = self.at("..=")
            self.i += 1
            hi = None
            if self._expr_start() and not (ns and self.at("{")):
                hi = self.parse_bin(0, ns)
            return N("range", lo=None, hi=hi, incl=incl)
        lo = self.parse_bin(0, ns, stmt)
        if self.at("..") or self.at("..="):
            incl = self.at("..=")
            self.i += 1
            hi = None
            if self._expr_start() and not (ns and self.at("{")):
                hi = self.parse_bin(0, ns)
            return N("range", lo=lo, hi=hi, incl=incl)
        return lo

    def parse_bin(self, minp, ns, stmt=False):
        lhs = self.parse_cast(ns, stmt)
        if stmt and lhs.k in BLOCK_LIKE and not self.at(".") and not self.at("?"):
            return lhs  # statement-position block-like expression ends here
        while True:
            tk = self.peek()
            if tk.k != "op" or tk.v not in BINPREC:
                break
            p = BINPREC[tk.v]
            if p < minp + 1 and not (p >= minp + 1):
                break
            if p <= minp:
                break
            self.i += 1
            rhs = self.parse_bin(p, ns)
            lhs = N("bin", op=tk.v, l=lhs, r=rhs)
        return lhs

    def parse_cast(self, ns, stmt=False):
        e = self.parse_unary(ns, stmt)
        while self.at("as"):
            self.i += 1
            e = N("cast", e=e, ty=self.parse_type())
        return e

    def parse_unary(self, ns=False, stmt=False):
        tk = self.peek()
        if tk.k == "op":
            if tk.v == "-":
                self.i += 1
                return N("unary", op="-", e=self.parse_unary(ns))
            if tk.v == "!":
                self.i += 1
                return N("unary", op="!", e=self.parse_unary(ns))
            if tk.v == "*":
                self.i += 1
                return N("deref", e=self.parse_unary(ns))
            if tk.v in ("&", "&&"):
                self.i += 1
                mut = self.eat("mut")
                if self.at("raw"):
                    self.err("&raw unsupported")
                inner = self.parse_unary(ns)
                e = N("ref", mut=mut, e=inner)
                if tk.v == "&&":
                    e = N("ref", mut=False, e=e)
                return e
        return self.parse_postfix(ns, stmt)

    def parse_postfix(self, ns, stmt=False):
        e = self.parse_primary(ns)
        if stmt and e.k in BLOCK_LIKE and not self.at(".") and not self.at("?"):
            return e
        while True:
            if self.at("?"):
                self.i += 1
                e = N("try", e=e)
            elif self.at("("):
                self.i += 1
                args = self.parse_args(")")
                e = N("call", f=e, args=args)
            elif self.at("["):
                self.i += 1
                idx = self.parse_expr()
                self.expect("]")
                e = N("index", e=e, i=idx)
            elif self.at("."):
                nx = self.peek(1)
                if nx.k == "int":
                    self.i += 2
                    e = N("field", e=e, name=str(nx.v[0]))
                elif nx.k == "float":
                    # x.0.1 lexed as float: split
                    self.i += 2
                    a, b = repr(nx.v[0]).split(".")
                    e = N("field", e=N("field", e=e, name=a), name=b)
                elif nx.k == "id":
                    if nx.v == "await":
                        self.err("await")
                    self.i += 2
                    name = nx.v
                    gen = None
                    if self.at("::") and self.at("<", 1):
                        self.i += 2
                        gen = self.parse_generic_args()
                    if self.at("("):
                        self.i += 1
                        args = self.parse_args(")")
                        e = N("mcall", recv=e, name=name, args=args, gen=gen, line=nx.line)
                    else:
                        e = N("field", e=e, name=name)
                else:
                    break
            else:
                break
        return e

    def parse_args(self, close):
        args = []
        while not self.at(close):
            args.append(self.parse_expr())
            if not self.eat(","):
                break
        self.expect(close)
        return args

    def parse_primary(self, ns):
        tk = self.peek()
        if tk.k == "int":
            self.i += 1
            return N("int", v=tk.v[0], suffix=tk.v[1])
        if tk.k == "float":
            self.i += 1
            return N("float", v=tk.v[0], suffix=tk.v[1])
        if tk.k == "str":
            self.i += 1
            return N("str", v=tk.v)
        if tk.k == "char":
            self.i += 1
            return N("char", v=tk.v)
        if tk.k == "life":  # labeled loop
            self.i += 1
            self.expect(":")
            e = self.parse_primary(ns)
            e.label = tk.v
            return e
        if tk.k == "op":
            if tk.v == "(":
                self.i += 1
                if self.eat(")"):
                    return N("tuple", els=[])
                first = self.parse_expr()
                if self.eat(")"):
                    return N("paren", e=first)
                els = [first]
                while self.eat(","):
                    if self.at(")"):
                        break
                    els.append(self.parse_expr())
                self.expect(")")
                return N("tuple", els=els)
            if tk.v == "[":
                self.i += 1
                if self.eat("]"):
                    return N("array", els=[])
                first = self.parse_expr()
                if self.eat(";"):
                    n = self.parse_expr()
                    self.expect("]")
                    return N("repeat", e=first, n=n)
                els = [first]
                while self.eat(","):
                    if self.at("]"):
                        break
                    els.append(self.parse_expr())
                self.expect("]")
                return N("array", els=els)
            if tk.v == "{":
                return self.parse_block()
            if tk.v in ("|", "||"):
                return self.parse_closure()
            if tk.v == "<":  # <T as Trait>::f
                self.i += 1
                t = self.parse_type()
                if self.eat("as"):
                    self.parse_type()
                self.split_shift()
                self.expect(">")
                segs = []
                while self.eat("::"):
                    segs.append(self.ident())
                return N("qpath", ty=t, rest=segs)
            if tk.v == "::":
                return N("pathx", path=self.parse_path())
        if tk.k == "id":
            v = tk.v
            if v in ("true", "false"):
                self.i += 1
                return N("bool", v=v == "true")
            if v == "if":
                return self.parse_if()
            if v == "match":
                self.i += 1
                scrut = self.parse_expr(no_struct=True)
                self.expect("{")
                arms = []
                while not self.at("}"):
                    self.skip_attrs()
                    pat = self.parse_pattern()
                    guard = None
                    if self.eat("if"):
                        guard = self.parse_expr()
                    self.expect("=>")
                    body = self.parse_expr(stmt=True)
                    arms.append((pat, guard, body))
                    if not self.eat(","):
                        if self.at("}"):
                            break
                        if body.k not in BLOCK_LIKE:
                            self.err("expected , after match arm")
                self.expect("}")
                return N("match", e=scrut, arms=arms)
            if v == "loop":
                self.i += 1
                return N("loop", body=self.parse_block(), label=None)
            if v == "while":
                self.i += 1
                if self.eat("let"):
                    pat = self.parse_pattern()
                    self.expect("=")
                    e = self.parse_expr(no_struct=True)
                    return N("while", let=(pat, e), cond=None, body=self.parse_block(), label=None)
                cond = self.parse_expr(no_struct=True)
                return N("while", let=None, cond=cond, body=self.parse_block(), label=None)
            if v == "for":
                self.i += 1
                pat = self.parse_pattern()
                self.expect("in")
                it = self.parse_expr(no_struct=True)
                return N("for", pat=pat, it=it, body=self.parse_block(), label=None)
            if v == "unsafe" and self.at("{", 1):
                self.i += 1
                b = self.parse_block()
                return N("unsafe", body=b)
            if v == "move" and (self.at("|", 1) or self.at("||", 1)):
                self.i += 1
                return self.parse_closure()
            if v == "return":
                self.i += 1
                e = None
                if self._expr_start() and not self.at("}"):
                    e = self.parse_expr()
                return N("return", e=e)
            if v == "break":
                self.i += 1
                label = None
                if self.peek().k == "life":
                    label = self.peek().v
                    self.i += 1
                e = None
                if self._expr_start() and not self.at("}") and not (ns and self.at("{")):
                    e = self.parse_expr()
                return N("break", e=e, label=label)
            if v == "continue":
                self.i += 1
                label = None
                if self.peek().k == "life":
                    label = self.peek().v
                    self.i += 1
                return N("continue", label=label)
            if v == "const" and self.at("{", 1):
                self.i += 1
                return self.parse_block()
            # macro call?
            if self.at("!", 1) and not self.at("!=", 1):
                return self.parse_macro()
            path = self.parse_path()
            if self.at("!") and len(path.segs) > 1 and not self.at("!=") and \
               (self.at("(", 1) or self.at("[", 1) or self.at("{", 1)):
                self.i -= 1
                self.t[self.i] = Tok("id", path.segs[-1], 0, tk.line)
                return self.parse_macro()
            if self.at("{") and not ns and self._looks_like_struct_lit():
                self.i += 1
                fields = []
                base = None
                while not self.at("}"):
                    self.skip_attrs()
                    if self.eat(".."):
                        base = self.parse_expr()
                        break
                    ftk = self.peek()
                    self.i += 1
                    fname = ftk.v if ftk.k == "id" else str(ftk.v[0])
                    if self.eat(":"):
                        fields.append((fname, self.parse_expr()))
                    else:
                        fields.append((fname, N("pathx", path=N("path", segs=[fname], gen={}))))
                    if not self.eat(","):
                        break
                self.expect("}")
                return N("structlit", path=path, fields=fields, base=base)
            return N("pathx", path=path)
        self.err("unexpected token in expression")

    def _looks_like_struct_lit(self):
        # `{ ident :` or `{ ident ,` or `{ ident }` or `{ .. ` or `{ }` (with a capitalised path)
        a, b = self.peek(1), self.peek(2)
        if a.k == "op" and a.v == "}":
            return True
        if a.k == "op" and a.v == "..":
            return True
        if a.k in ("id", "int") and b.k == "op" and b.v in (":", ",", "}"):
            if b.v == ":" and self.at(":", 3):
                return False
            return True
        return False

    def parse_if(self):
        self.expect("if")
        if self.eat("let"):
            pat = self.parse_pattern()
            self.expect("=")
            e = self.parse_expr(no_struct=True)
            cond = N("iflet", pat=pat, e=e)
        else:
            cond = self.parse_expr(no_struct=True)
        then = self.parse_block()
        els = None
        if self.eat("else"):
            if self.at("if"):
                els = self.parse_if()
            else:
                els = self.parse_block()
        return N("if", cond=cond, then=then, els=els)

    def parse_closure(self):
        params = []
        if not self.eat("||"):
            self.expect("|")
            while not self.at("|"):
                pat = self.parse_pattern1()
                ty = None
                if self.eat(":"):
                    ty = self.parse_type()
                params.append((pat, ty))
                if not self.eat(","):
                    break
            self.expect("|")
        if self.eat("->"):
            self.parse_type()
            body = self.parse_block()
        else:
            body = self.parse_expr()
        return N("closure", params=params, body=body)

    def parse_macro(self):
        name = self.ident()
        self.expect("!")
        tk = self.peek()
        close = {"(": ")", "[": "]", "{": "}"}[tk.v]
        brace = tk.v == "{"
        start = self.i
        if name in ("assert", "debug_assert", "assert_eq", "debug_assert_eq", "assert_ne", "debug_assert_ne",
                    "izip", "vec", "panic", "unreachable", "unimplemented", "todo", "println", "eprintln",
                    "format", "write", "writeln", "print", "min", "max"):
            self.i += 1
            args = []
            rep = None
            while not self.at(close):
                args.append(self.parse_expr())
                if name == "vec" and self.eat(";"):
                    rep = self.parse_expr()
                    break
                if not self.eat(","):
                    break
            self.expect(close)
            return N("macro", name=name, args=args, rep=rep, brace=brace)
        if name == "matches":
            self.i += 1
            e = self.parse_expr()
            self.expect(",")
            pat = self.parse_pattern()
            guard = None
            if self.eat("if"):
                guard = self.parse_expr()
            self.eat(",")
            self.expect(close)
            return N("macro", name=name, args=[e], pat=pat, guard=guard, brace=brace, rep=None)
        # unknown macro: keep the raw token range
        self.skip_balanced(tk.v, close)
        return N("macro", name=name, args=None, raw=self.t[start:self.i], brace=brace, rep=None)


def parse_file(path):
    src = open(path).read()
    p = Parser(lex(src), path)
    items = p.parse_items()
    return p, items
