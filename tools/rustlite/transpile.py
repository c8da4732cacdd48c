"""Rust-subset AST -> Python source (test infrastructure, see __init__.py).

`Crate` loads reference .rs files, keeps a registry of their items and
compiles functions to Python on first use.  Every compiled function takes a
leading `_g` dict with the generic bindings in force ({"T": "u8", "N": 4}).
"""
import keyword
import os

from . import runtime as R
from .parser import N, Parser, parse_file

INT = R.INT_BITS
PRIMS = set(INT) | {"f32", "f64", "bool", "char", "str"}
RESERVED = set(keyword.kwlist) | {
    "min", "max", "abs", "len", "range", "int", "list", "tuple", "type", "iter", "next", "map", "zip", "sum",
    "any", "all", "filter", "bool", "float", "print", "id", "dir", "vars", "input", "object", "set", "dict",
    "str", "repr", "hash", "pow", "round", "slice", "sorted", "reversed", "enumerate", "format", "bytes",
    "self_", "None", "True", "False", "R", "N",
}
NONZERO = {"NonZeroU8": "u8", "NonZeroU16": "u16", "NonZeroU32": "u32", "NonZeroU64": "u64",
           "NonZeroUsize": "usize"}
WRAPPERS = ("MaybeUninit", "Box", "Aligned", "Arc", "Rc", "Cow", "ManuallyDrop", "RefCell", "Cell", "Mutex",
            "RwLock")
VECS = ("Vec", "ArrayVec", "VecDeque", "SmallVec")


class TranspileError(Exception):
    pass


def pyfield(name):
    if name[0].isdigit():
        return "_" + name
    if keyword.iskeyword(name) or name in ("get", "set"):
        return name + "_"
    return name


def fits(src, dst):
    """every value of int type src representable in dst?"""
    if src not in INT or dst not in INT:
        return False
    return R.int_min(dst) <= R.int_min(src) and R.int_max(src) <= R.int_max(dst)


class FnInfo:
    def __init__(self, node, parser, modpath, fname, owner=None, owner_gens=(), trait=None):
        self.node, self.parser, self.modpath, self.fname = node, parser, modpath, fname
        self.owner, self.owner_gens, self.trait = owner, list(owner_gens), trait
        self.pyname = None
        self.compiled = False
        self.source = None

    @property
    def gens(self):
        return self.owner_gens + [g for g in self.node.gens]


class StructInfo:
    def __init__(self, node, cls, ftypes):
        self.node, self.cls, self.ftypes = node, cls, ftypes


class Crate:
    def __init__(self, root="/root/reference/src", checks=True, strict=False):
        # strict: also range-check UNTYPED `let`s against their inferred integer type (a debug
        # build of the reference checks every arithmetic result; with this on, a case that runs
        # to the end overflowed nowhere a `let` could see, so debug == release for it)
        self.strict = strict
        self.root = root
        self.checks = checks
        self.files = {}
        self.fns = {}        # name -> [FnInfo]
        self.methods = {}    # type -> {name: [FnInfo]}
        self.consts = {}     # name or Type::name -> [(node, modpath, fname, owner, parser)]
        self.structs = {}
        self.enums = {}      # name -> node
        self.variants = {}   # variant -> [enum name]
        self.aliases = {}
        self.traits = {}     # trait name -> {method: FnInfo}  (default methods)
        self.trait_impls = {}  # type -> [trait names]
        self.G = {"R": R, "_g0": {}}
        self.G.update({
            "_cast": R.cast, "_chk": R.chk, "_div": R.div, "_rem": R.rem, "_deref": R.deref, "_store": R.store,
            "_Cell": R.Cell, "_RRef": R.RRef, "_refl": R.refmut_local, "_refi": R.refmut_index,
            "_reff": R.refmut_field, "_Rg": R.RRange, "_S": R.RSlice, "_repeat": R.repeat, "_array": R.array,
            "_cp": R.copy_if_array, "_Some": R.Some, "_NONE": R.NONE, "_Ok": R.Ok, "_Err": R.Err,
            "_Panic": R.Panic, "_izip": R.izip, "_ii": R.into_iter, "_RIter": R.RIter, "_im": R.int_method,
            "_msb": R.msb, "_round_shift": R.round_shift, "_clamp": R.clamp3, "_E": R.REnum,
            "_mc": self._mc, "_tf": self._tf, "_fv": self._fv, "_not": self._not, "_try_from": R.try_from,
            "_RPtr": R.RPtr, "_raw": R.slice_from_raw_parts, "_vrep": self._vec_repeat, "_swap": self._swap,
            "_with": self._with,
        })
        self._uniq = 0
        self._index = None
        self.const_cache = {}
        self.enum_vals = {}
        self.shims = {}  # type name -> python class (v_frame stand-ins, etc.)
        for cls in (R.Plane, R.PlaneSlice, R.PlaneRegion, R.PlaneConfig, R.PlaneOffset, R.Rect):
            self.shims[cls._rname] = cls
            cls._crate = self
        self.shims["PlaneRegionMut"] = R.PlaneRegion
        self.shims["PlaneMutSlice"] = R.PlaneSlice

    # ------------------------------------------------------------ loading
    def autoload(self, name):
        """an unresolved name: find the reference file that defines it and load that"""
        import re
        if self._index is None:
            self._index = {}
            for dp, dn, fn in os.walk(self.root):
                rel = os.path.relpath(dp, self.root)
                if rel.split(os.sep)[0] in ("asm", "bin", "test_encode_decode", "capi"):
                    continue
                for f in sorted(fn):
                    if not f.endswith(".rs") or f in ("test.rs", "capi.rs", "fuzzing.rs"):
                        continue
                    r = os.path.normpath(os.path.join(rel, f))
                    txt = open(os.path.join(dp, f)).read()
                    for m in re.finditer(r"\b(?:const|static|fn|struct|enum|type|trait)\s+([A-Za-z_][A-Za-z0-9_]*)", txt):
                        self._index.setdefault(m.group(1), []).append(r)
        for r in self._index.get(name, ()):
            if r not in self.files:
                self.load(r)
                return True
        return False

    def load(self, rel, tests=False):
        """tests=True also registers the file's `#[cfg(test)]` items: the reference's own unit
        tests, which tests/test_reference_unit_tests.py runs as this transpiler's self-test"""
        if rel in self.files:
            return
        p, items = parse_file(os.path.join(self.root, rel))
        self.files[rel] = (p, items)
        self.include_tests = tests
        try:
            self._register(items, p, (), rel)
        finally:
            self.include_tests = False

    def define_py(self, name, pyfunc):
        """A crate-level function supplied by the generator as a Python callable f(_g, *args): for
        the few reference functions that live inside a macro body (macro expansion is not part of
        this transpiler), e.g. get_func of impl_1d_tx! (src/transform/forward_shared.rs:201-218).
        The generator says what stands behind it."""
        node = N("fn", name=name, gens=[], params=[], ret=None, body=None, has_self=False, attrs=[], parsed=None)
        info = FnInfo(node, None, (), "<python: %s>" % name)
        info.pyname = "py_%s_%d" % (name, len(self.G))
        info.compiled = True
        self.G[info.pyname] = pyfunc
        self.fns.setdefault(name, []).insert(0, info)

    def load_text(self, name, text):
        """Rust source text that is not a file of the tree: a function wrapped around a SLICE of a
        reference function's lines (the generator states which lines, and what the wrapper
        replaces), parsed and registered like a file."""
        from .lexer import lex
        from .parser import Parser
        if name in self.files:
            return
        p = Parser(lex(text), name)
        items = p.parse_items()
        self.files[name] = (p, items)
        self._register(items, p, (), name)

    def _register(self, items, p, modpath, fname, owner=None, owner_gens=(), trait=None):
        for it in items:
            if any("cfg ( test )" in a or "cfg(test)" in a.replace(" ", "") for a in it.attrs) and \
                    not getattr(self, "include_tests", False):
                continue
            k = it.k
            if k == "mod":
                self._register(it.items, p, modpath + (it.name,), fname)
            elif k == "fn":
                info = FnInfo(it, p, modpath, fname, owner, owner_gens, trait)
                if owner:
                    self.methods.setdefault(owner, {}).setdefault(it.name, []).append(info)
                else:
                    self.fns.setdefault(it.name, []).append(info)
            elif k == "const":
                key = (owner + "::" + it.name) if owner else it.name
                self.consts.setdefault(key, []).append((it, modpath, fname, owner, p))
            elif k == "struct":
                if it.name in self.shims:
                    continue
                self._make_struct(it)
            elif k == "enum":
                self.enums[it.name] = it
                d = 0
                for (vn, kind, fields, disc) in it.variants:
                    self.variants.setdefault(vn, []).append(it.name)
            elif k == "alias":
                if it.ty is not None:
                    self.aliases[it.name] = it.ty
            elif k == "impl":
                tn = self._type_name(it.ty)
                if tn is None:
                    continue
                tr = self._type_name(it.trait) if it.trait is not None else None
                if tr:
                    self.trait_impls.setdefault(tn, []).append(tr)
                self._register(it.items, p, modpath, fname, owner=tn, owner_gens=it.gens, trait=tr)
            elif k == "trait":
                d = self.traits.setdefault(it.name, {})
                for sub in it.items:
                    if sub.k == "fn" and sub.body is not None:
                        d[sub.name] = FnInfo(sub, p, modpath, fname, owner="<trait %s>" % it.name,
                                             owner_gens=it.gens, trait=it.name)

    def _type_name(self, t):
        if t is None:
            return None
        if t.k == "path":
            return t.segs[-1]
        if t.k in ("tref", "tptr"):
            return self._type_name(t.inner)
        if t.k == "tslice" or t.k == "tarray":
            return "[]"
        return None

    def _make_struct(self, node):
        fields = tuple(pyfield(f) for f, _ in node.fields)
        is_copy = any("derive" in a and "Copy" in a for a in node.attrs)
        cls = type("S_" + node.name, (R.RStruct,), {"_fields": fields, "_is_copy": is_copy,
                                                     "_rname": node.name, "_crate": self})
        self.structs[node.name] = StructInfo(node, cls, {f: t for f, t in node.fields})
        self.G["S_" + node.name] = cls

    # ------------------------------------------------------------ runtime helpers bound to the crate
    def _mc(self, g, recv, hint, name, args):
        """method call with run-time dispatch"""
        t = type(recv)
        if t is int or t is bool or t is float:
            ms = self.methods.get(hint) if hint else None
            if ms and name in ms:
                return self._call_info(ms[name][0], g, (recv,) + args)
            if hint is None or hint not in self.methods:
                info = self._prim_trait_method(name)
                if info is not None and name not in ("min", "max", "abs", "clamp", "pow"):
                    return self._call_info(info, g, (recv,) + args)
            return R.int_method(recv, hint if hint in INT else None, name, args)
        if t is tuple:
            return R.tuple_method(recv, name, args)
        if issubclass(t, list):   # a raw backing store (Plane::data): the slice methods on a view of it
            return self._mc(g, R.RSlice(recv), hint, name, args)
        if t is R.Cell or t is R.RRef:
            if name == "write" and len(args) == 1:
                recv.set(args[0])
                return None
            return self._mc(g, recv.get(), hint, name, args)
        rn = getattr(t, "_rname", None)
        if rn is not None:
            ms = self.methods.get(rn)
            if ms and name in ms:
                return self._call_info(ms[name][0], g, (recv,) + args)
            for tr in self.trait_impls.get(rn, ()):
                d = self.traits.get(tr)
                if d and name in d:
                    return self._call_info(d[name], g, (recv,) + args)
        if t is R.REnum:
            ms = self.methods.get(recv.ty)
            if ms and name in ms:
                return self._call_info(ms[name][0], g, (recv,) + args)
            if name == "or":
                name = "or_"
        f = getattr(recv, name, None)
        if f is None:
            if name in ("clone", "to_owned", "borrow", "as_ref", "as_mut", "into", "deref", "by_ref", "as_const"):
                return recv
            ms = self.methods.get(rn) if rn is not None else None
            if ms and "next" in ms and hasattr(R.RIter, name):
                # a struct that implements Iterator (its own `next`): the provided adaptors
                # (map, zip, take, ...) run on a stream of its next() results
                nxt = ms["next"][0]

                def stream():
                    while True:
                        v = self._call_info(nxt, g, (recv,))
                        if not (isinstance(v, R.REnum) and v.var == "Some"):
                            return
                        yield v.p[0]
                return getattr(R.RIter(stream()), name)(*args)
            raise R.Panic("rustlite: no method %s on %r" % (name, t.__name__))
        return f(*args)

    def _prim_trait_method(self, name):
        for tn in ("i32", "i16", "i64", "u32", "u16", "u8", "usize"):
            ms = self.methods.get(tn)
            if ms and name in ms:
                return ms[name][0]
        return None

    def call_method(self, recv, name, args):
        return self._mc({}, recv, None, name, tuple(args))

    def _call_info(self, info, g, args):
        # run-time dispatched (method) calls hand the caller's bindings on by name as well: a
        # pixel type arriving at a `T: Coefficient` parameter is the caller's `T: Pixel` (see
        # gen_dict) -- a Coefficient is never u8 / u16
        for gg in info.gens:
            if gg[0] == "type" and gg[2] and "Coefficient" in gg[2] and g.get(gg[1]) in ("u8", "u16"):
                g = dict(g)
                g[gg[1]] = "i16" if g[gg[1]] == "u8" else "i32"
        return self.pyfn(info)(g, *args)

    def _tf(self, v, i):
        if type(v) is tuple:
            return v[i]
        if type(v) in (R.Cell, R.RRef):
            return self._tf(v.get(), i)
        return getattr(v, "_%d" % i)

    def _fv(self, f, g):
        import functools
        return functools.partial(f, g)

    def _not(self, v):
        return (not v) if type(v) is bool else ~v

    def _vec_repeat(self, v, n):
        s = R.repeat(v, n)
        s.arr = False
        return s

    def _swap(self, a, b):
        va, vb = R.deref(a), R.deref(b)
        if isinstance(a, (R.Cell, R.RRef)):
            a.set(vb)
            b.set(va)
        else:
            raise R.Panic("mem::swap of non-scalar places")

    def _with(self, base, **kw):
        o = base._copy()
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    # ------------------------------------------------------------ lookup
    def find_fn(self, name, ctx=None):
        c = self.fns.get(name)
        if not c:
            return None
        if len(c) == 1 or ctx is None:
            return c[0]
        best = [i for i in c if i.fname == ctx.fname and i.modpath == ctx.modpath]
        if best:
            return best[0]
        best = [i for i in c if i.fname == ctx.fname and ctx.modpath[:len(i.modpath)] == i.modpath]
        if best:
            return sorted(best, key=lambda i: -len(i.modpath))[0]
        best = [i for i in c if i.fname == ctx.fname]
        if best:
            return best[0]
        return c[0]

    def find_method(self, ty, name):
        ms = self.methods.get(ty)
        if ms and name in ms:
            return ms[name][0]
        for tr in self.trait_impls.get(ty, ()):
            d = self.traits.get(tr)
            if d and name in d:
                return d[name]
        return None

    def pyfn(self, info):
        if not info.compiled:
            FnCompiler(self, info).compile()
        return self.G.get(info.pyname)  # None while a recursive function is being compiled

    def get(self, name, owner=None, file=None):
        """Python callable f(_g, *args) for a reference function."""
        if owner:
            info = self.find_method(owner, name)
        else:
            c = self.fns.get(name) or []
            if file:
                c = [i for i in c if i.fname == file] or c
            info = c[0] if c else None
        if info is None:
            raise KeyError(name)
        return self.pyfn(info)

    def const(self, name, ctx=None):
        c = self.consts.get(name)
        if not c:
            raise KeyError(name)
        ent = c[0]
        if ctx is not None and len(c) > 1:
            same = [e for e in c if e[2] == ctx.fname and e[1] == ctx.modpath] or \
                   [e for e in c if e[2] == ctx.fname]
            if same:
                ent = same[0]
        key = (name, ent[2], ent[1])
        if key not in self.const_cache:
            node, modpath, fname, owner, parser = ent
            pyname = "K_%s_%d" % (name.replace("::", "__"), len(self.const_cache))
            fake = FnInfo(N("fn", name="<const %s>" % name, gens=[], params=[], ret=node.ty, body=None,
                            has_self=False, attrs=[], parsed=None), parser, modpath, fname, owner)
            fc = FnCompiler(self, fake)
            self.const_cache[key] = pyname  # break cycles
            self.G[pyname] = fc.eval_const(node.init, node.ty)
        return self.const_cache[key]

    def const_value(self, name):
        return self.G[self.const(name)]

    def enum_value(self, ename, vname):
        key = (ename, vname)
        if key not in self.enum_vals:
            node = self.enums[ename]
            d = 0
            for (vn, kind, fields, disc) in node.variants:
                if disc is not None:
                    fake = FnInfo(N("fn", name="<disc>", gens=[], params=[], ret=None, body=None, has_self=False,
                                    attrs=[], parsed=None), None, (), "?", None)
                    d = FnCompiler(self, fake).eval_const(disc, None)
                if kind == "unit":
                    pn = "E_%s__%s" % (ename, vn)
                    self.G[pn] = R.REnum(ename, vn, d)
                    self.enum_vals[(ename, vn)] = (pn, d, kind, fields)
                else:
                    self.enum_vals[(ename, vn)] = (None, d, kind, fields)
                d += 1
        return self.enum_vals[key]

    def define_enum(self, name, variants):
        """an enum of a third-party crate (e.g. v_frame's ChromaSampling): unit variants"""
        node = N("enum", name=name, variants=[(v, "unit", [], None) for v in variants], attrs=[])
        self.enums[name] = node
        for v in variants:
            self.variants.setdefault(v, []).append(name)

    def uniq(self, base):
        self._uniq += 1
        return "%s_%d" % (base, self._uniq)

    def dump(self):
        out = []
        for lst in list(self.fns.values()) + [l for m in self.methods.values() for l in m.values()]:
            for info in lst:
                if info.source:
                    out.append(info.source)
        return "\n\n".join(out)


# ====================================================================== function compiler
class Frame:
    def __init__(self):
        self.declared = set()
        self.assigned_outer = set()


class FnCompiler:
    def __init__(self, crate, info):
        self.c = crate
        self.info = info
        self.lines = []
        self.ind = 1
        self.scopes = [{}]
        self.counts = {}
        self.tmpn = 0
        self.boxed = set()
        self.frames = [Frame()]
        self.loops = []  # (label, kind, flags dict)
        self.type_params = set()
        self.const_params = set()
        for g in info.gens:
            (self.type_params if g[0] == "type" else self.const_params).add(g[1])
        self.local_items = [{}]
        self.ret_ty = None
        self.flagnames = set()

    # ------------------------------------------------------------ output helpers
    def emit(self, s):
        self.lines.append("    " * self.ind + s)

    def tmp(self, base="_t"):
        self.tmpn += 1
        n = "%s%d" % (base, self.tmpn)
        self.frames[-1].declared.add(n)
        return n

    def capture(self, fn):
        save = self.lines
        self.lines = []
        try:
            r = fn()
        finally:
            sub = self.lines
            self.lines = save
        return sub, r

    def err(self, msg, node=None):
        raise TranspileError("%s::%s: %s" % (self.info.fname, self.info.node.name, msg))

    # ------------------------------------------------------------ scopes / variables
    def push(self):
        self.scopes.append({})
        self.local_items.append({})

    def pop(self):
        self.scopes.pop()
        self.local_items.pop()

    def lookup(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s[name]
        return None

    def lookup_item(self, name):
        for s in reversed(self.local_items):
            if name in s:
                return s[name]
        return None

    def declare(self, name, ty=None):
        n = self.counts.get(name, 0) + 1
        self.counts[name] = n
        py = name if n == 1 else "%s_%d" % (name, n)
        if py in RESERVED or py.startswith("_") and py[1:2].isalpha() and py in self.c.G:
            py = py + "_v"
        if py in ("self",):
            py = "self_"
        ent = {"py": py, "ty": ty, "boxed": name in self.boxed}
        self.scopes[-1][name] = ent
        self.frames[-1].declared.add(py)
        return ent

    def assign_var(self, ent, val):
        if ent["boxed"]:
            self.emit("%s.v = %s" % (ent["py"], val))
        else:
            if ent["py"] not in self.frames[-1].declared:
                self.frames[-1].assigned_outer.add(ent["py"])
            self.emit("%s = %s" % (ent["py"], val))

    def init_var(self, name, val, ty=None):
        ent = self.declare(name, ty)
        if ent["boxed"]:
            self.emit("%s = _Cell(%s)" % (ent["py"], val))
        else:
            self.emit("%s = %s" % (ent["py"], val))
        return ent

    # ------------------------------------------------------------ types
    def norm(self, t):
        """type node -> static type: 'u32' | 'Name' | ('arr', el) | ('tup', [..]) | ('ref', x) | ('opt', x) | None"""
        if t is None:
            return None
        k = t.k
        if k == "tref":
            inner = self.norm(t.inner)
            return ("ref", inner, bool(t.mut))
        if k == "tptr":
            return ("ptr", self.norm(t.inner))
        if k == "tarray":
            return ("arr", self.norm(t.el), t.n)   # the length expression rides along (default_for)
        if k == "tslice":
            return ("arr", self.norm(t.el))
        if k == "ttuple":
            return ("tup", [self.norm(e) for e in t.els])
        if k == "path":
            name = t.segs[-1]
            ga = t.gen.get(len(t.segs) - 1, [])
            if name == "Self" and self.info.owner:
                return self.info.owner
            if name in NONZERO:
                return NONZERO[name]
            if name in WRAPPERS and ga:
                return self.norm(ga[0]) if ga[0].k != "gconst" else None
            if name in VECS and ga:
                return ("arr", self.norm(ga[0]))
            if name == "Option" and ga:
                return ("opt", self.norm(ga[0]))
            if name in self.c.aliases and name not in self.c.structs:
                return self.norm(self.c.aliases[name])
            return name
        return None

    @staticmethod
    def strip(t):
        while isinstance(t, tuple) and t[0] == "ref":
            t = t[1]
        return t

    def tyname(self, t):
        """static type -> expression giving the run-time type name (or None)"""
        t = self.strip(t)
        if isinstance(t, str):
            if t in self.type_params:
                return "_g[%r]" % t
            if t in INT or t in ("f32", "f64", "bool"):
                return repr(t)
            return repr(t)
        return None

    def ty(self, e):
        """static type inference (best effort; None = unknown)"""
        k = e.k
        if k == "int":
            return e.suffix
        if k == "float":
            return e.suffix or "f64"
        if k == "bool":
            return "bool"
        if k == "paren":
            return self.ty(e.e)
        if k == "cast":
            return self.norm(e.ty)
        if k == "pathx":
            p = e.path
            if len(p.segs) == 1:
                ent = self.lookup(p.segs[0])
                if ent is not None:
                    return ent["ty"]
                if p.segs[0] in self.const_params:
                    return "usize"
                c = self.c.consts.get(p.segs[0])
                if c:
                    return self.norm(c[0][0].ty)
                return None
            if len(p.segs) == 2:
                a, b = p.segs
                if a == "Self" and self.info.owner:
                    a = self.info.owner
                if a in INT and b in ("MAX", "MIN", "BITS"):
                    return a if b != "BITS" else "u32"
                c = self.c.consts.get(a + "::" + b)
                if c:
                    return self.norm(c[0][0].ty)
                if a in self.c.enums:
                    return a
            return None
        if k == "unary":
            return self.ty(e.e)
        if k == "deref":
            t = self.ty(e.e)
            if isinstance(t, tuple) and t[0] in ("ref", "ptr"):
                return t[1]
            return t
        if k == "ref":
            t = self.ty(e.e)
            return ("ref", t) if t is not None else None
        if k == "bin":
            if e.op in ("==", "!=", "<", ">", "<=", ">=", "&&", "||"):
                return "bool"
            lt = self.strip(self.ty(e.l))
            if e.op in ("<<", ">>"):
                return lt
            if lt is not None:
                return lt
            return self.strip(self.ty(e.r))
        if k == "index":
            t = self.strip(self.ty(e.e))
            if isinstance(t, tuple) and t[0] in ("arr", "ptr"):
                if e.i.k == "range":
                    return t
                return t[1]
            if t == "PlaneRegion" or t == "PlaneRegionMut" or t == "PlaneSlice":
                return ("arr", None)
            return None
        if k == "field":
            t = self.strip(self.ty(e.e))
            if isinstance(t, tuple) and t[0] == "tup" and e.name.isdigit() and int(e.name) < len(t[1]):
                return t[1][int(e.name)]
            if isinstance(t, str) and t in self.c.structs:
                ft = self.c.structs[t].ftypes.get(e.name)
                if ft is not None:
                    save = self.info.owner
                    return self.norm(ft)
            return None
        if k == "call":
            f = e.f
            if f.k == "pathx":
                segs = f.path.segs
                if len(segs) == 2 and segs[1] in ("cast_from", "from", "try_from", "new", "zero", "one") and \
                        (segs[0] in INT or segs[0] in self.type_params or segs[0] in ("f64", "f32")):
                    return segs[0] if segs[1] != "try_from" else None
                if len(segs) == 2 and segs[0] in NONZERO:
                    return None
                info = self.resolve_fn(f.path)
                if info is not None and info.node.ret is not None:
                    sub = FnCompiler(self.c, info)
                    return sub.norm(info.node.ret)
                if len(segs) == 1 and segs[0] in self.c.structs:
                    return segs[0]
                if segs[-1] in ("max", "min") and e.args:
                    return self.strip(self.ty(e.args[0]))
                if segs[-1] in ("msb",):
                    return "i32"
                if segs[-1] in ("round_shift", "clamp") and e.args:
                    return self.strip(self.ty(e.args[0]))
            return None
        if k == "mcall":
            rt = self.strip(self.ty(e.recv))
            n = e.name
            if n in ("min", "max", "clamp", "abs", "pow", "wrapping_add", "wrapping_sub", "wrapping_mul",
                     "wrapping_neg", "saturating_add", "saturating_sub", "saturating_mul", "get", "clone",
                     "rotate_left", "swap_bytes", "signum", "div_ceil", "next_power_of_two", "wrapping_shl",
                     "wrapping_shr", "rem_euclid", "div_euclid", "align_power_of_two",
                     "align_power_of_two_and_shift", "to_owned", "copied", "isqrt"):
                return rt
            if n in ("unwrap", "expect", "unwrap_or", "unwrap_or_default", "unwrap_unchecked") and \
                    isinstance(rt, tuple) and rt[0] == "opt":
                return rt[1]                                   # Option<T>::unwrap() -> T
            if n in ("get_unchecked", "get_unchecked_mut") and isinstance(rt, tuple) and rt[0] == "arr" and \
                    e.args and self.strip(self.ty(e.args[0])) in INT:
                return ("ref", rt[1], n.endswith("mut"))     # slice.get_unchecked(i) -> &T
            if n in ("unsigned_abs", "abs_diff"):
                if isinstance(rt, str) and rt in INT:
                    return "u" + rt[1:] if rt[0] == "i" else rt
                return None
            if n in ("len", "count", "trailing_zeros", "leading_zeros", "count_ones", "ilog2", "ilog"):
                return "usize" if n in ("len", "count") else "u32"
            if n == "into" or n == "as_":
                return None
            if isinstance(rt, str):
                info = self.c.find_method(rt, n)
                if info is not None and info.node.ret is not None:
                    return FnCompiler(self.c, info).norm(info.node.ret)
            if n in ("sum", "product") and e.gen:
                return self.norm(e.gen[0]) if e.gen[0].k != "gconst" else None
            return None
        if k == "if":
            t = self.ty_block(e.then)
            if t is None and e.els is not None:
                t = self.ty_block(e.els) if e.els.k == "block" else self.ty(e.els)
            return t
        if k == "block" or k == "unsafe":
            return self.ty_block(e if k == "block" else e.body)
        if k == "tuple":
            return ("tup", [self.ty(x) for x in e.els])
        if k == "repeat":
            return ("arr", self.ty(e.e))
        if k == "array":
            return ("arr", self.ty(e.els[0]) if e.els else None)
        if k == "structlit":
            n = e.path.segs[-1]
            return self.info.owner if n == "Self" else n
        if k == "macro" and e.name == "vec":
            return ("arr", self.ty(e.args[0]) if e.args else None)
        return None

    def ty_block(self, b):
        if b.k != "block" or b.tail is None or b.stmts:
            return None
        return self.ty(b.tail)

    def is_scalar_ty(self, t):
        t = self.strip(t)
        return isinstance(t, str) and (t in INT or t in ("f32", "f64", "bool"))

    # ------------------------------------------------------------ entry points
    def compile(self):
        info = self.info
        node = info.node
        c = self.c
        base = "_".join(info.modpath + ((info.owner,) if info.owner else ()) + (node.name,))
        base = "".join(ch if ch.isalnum() or ch == "_" else "_" for ch in base)
        info.pyname = c.uniq("f_" + base)
        info.compiled = True  # recursion guard
        if node.body is None:
            self.err("function has no body")
        if node.parsed is None:
            node.parsed = info.parser.parse_body(node.body)
        body = node.parsed
        self.prescan(body)
        params = ["_g"]
        pre = []
        if node.has_self:
            ent = self.declare("self", info.owner)
            params.append(ent["py"])
        for i, (pat, ty) in enumerate(node.params):
            nt = self.norm(ty)
            if pat.k in ("pident", "pbind") and not getattr(pat, "sub", None):
                name = pat.name
                ent = self.declare(name, nt)
                params.append(ent["py"])
                if ent["boxed"]:
                    pre.append("%s = _Cell(%s)" % (ent["py"], ent["py"]))
            else:
                a = "_a%d" % i
                params.append(a)
                pre.append((pat, a, nt))
        self.ret_ty = self.norm(node.ret)
        for i, (pat, ty) in enumerate(node.params):
            t = ty
            while t.k == "tref":
                t = t.inner
            if t.k == "tarray" and t.n.k == "pathx" and len(t.n.path.segs) == 1 and \
                    t.n.path.segs[0] in self.const_params:
                # a const generic that is inferred from an array argument's length
                self.emit("_g = {**_g, %r: len(%s)}" % (t.n.path.segs[0], params[i + 1 + int(node.has_self)]))
        for p in pre:
            if isinstance(p, str):
                self.emit(p)
            else:
                self.bind(p[0], p[1], p[2])
        val = self.block_value(body, newscope=False)
        if val is not None:
            self.emit("return " + self.ret_wrap(val))
        src = "def %s(%s):\n" % (info.pyname, ", ".join(params))
        src += "    # %s : %s%s\n" % (info.fname, "::".join(info.modpath + ((info.owner,) if info.owner else ())
                                                           + (node.name,)), "")
        for fl in sorted(self.flagnames):
            src += "    %s = False\n" % fl
        src += "\n".join(self.lines) if self.lines else "    pass"
        info.source = src
        try:
            exec(compile(src, "<rustlite %s>" % info.pyname, "exec"), c.G)
        except SyntaxError as ex:
            raise TranspileError("generated code does not compile for %s: %s\n%s" % (node.name, ex, src))
        return src

    def ret_wrap(self, val):
        t = self.strip(self.ret_ty)
        if self.c.checks and isinstance(t, str) and t in INT:
            return "_chk(%s, %r, 'return of %s')" % (val, t, self.info.node.name)
        return val

    def eval_const(self, init, ty):
        nt = self.norm(ty) if ty is not None else None
        self.expected = nt
        val = self.ex(init)
        src = "def _constfn(_g):\n" + "\n".join(self.lines) + "\n    return %s\n" % val
        g = self.c.G
        try:
            exec(compile(src, "<rustlite const>", "exec"), g)
        except SyntaxError as ex:
            raise TranspileError("const does not compile: %s\n%s" % (ex, src))
        return g["_constfn"]({})

    def prescan(self, node):
        """find locals whose address is taken with &mut (they are boxed)"""
        def walk(x):
            if isinstance(x, N):
                if x.k == "ref" and x.mut and x.e.k == "pathx" and len(x.e.path.segs) == 1:
                    self.boxed.add(x.e.path.segs[0])
                if x.k == "fn":
                    return
                for v in x.__dict__.values():
                    walk(v)
            elif isinstance(x, (list, tuple)):
                for v in x:
                    walk(v)
        walk(node)
        self.boxed.discard("self")

    # ------------------------------------------------------------ blocks and statements
    def block_value(self, b, newscope=True):
        """emit the statements; return the tail expression string (or None)"""
        if newscope:
            self.push()
        for s in b.stmts:
            self.stmt(s)
        val = None
        if b.tail is not None:
            val = self.ex(b.tail)
            if b.tail.k in ("pathx", "field", "index") and newscope:
                # value leaves the scope of its locals: materialise it
                t = self.tmp()
                self.emit("%s = %s" % (t, val))
                val = t
        if newscope:
            self.pop()
        return val

    def block_stmts(self, b):
        """statement context: value discarded"""
        n0 = len(self.lines)
        self.push()
        for s in b.stmts:
            self.stmt(s)
        if b.tail is not None:
            self.stmt(N("expr", e=b.tail, line=0))
        self.pop()
        if len(self.lines) == n0:
            self.emit("pass")

    def stmt(self, s):
        k = s.k
        if k == "let":
            self.let(s)
        elif k == "item":
            it = s.item
            if it.k == "fn":
                info = FnInfo(it, self.info.node.parsed.parser, self.info.modpath + ("<%s>" % self.info.node.name,),
                              self.info.fname, None, self.info.gens)
                self.local_items[-1][it.name] = ("fn", info)
            elif it.k == "const":
                pyname = self.c.uniq("K_local_" + it.name)
                fake = FnInfo(N("fn", name="<const>", gens=[], params=[], ret=it.ty, body=None, has_self=False,
                                attrs=[], parsed=None), self.info.parser, self.info.modpath, self.info.fname,
                              self.info.owner)
                fc = FnCompiler(self.c, fake)
                fc.local_items = [dict(d) for d in self.local_items]
                self.local_items[-1][it.name] = ("const", pyname, self.norm(it.ty))
                self.c.G[pyname] = fc.eval_const(it.init, it.ty)
            elif it.k == "struct":
                self.c._make_struct(it)
            # use / others ignored
        elif k == "expr":
            self.expr_stmt(s.e)
        else:
            self.err("stmt %s" % k)

    def let(self, s):
        nt = self.norm(s.ty) if s.ty is not None else None
        if s.init is None:
            # declared, assigned later
            self._declare_pat_names(s.pat, nt)
            return
        self.expected = nt
        val = self.ex(s.init)
        self.expected = None
        if nt is None:
            nt = self.ty(s.init)
        if s.els is not None:
            cond, binds = self.pat_cond(s.pat, self.simple(val))
            self.emit("if not (%s):" % cond)
            self.ind += 1
            self.block_stmts(s.els)
            self.ind -= 1
            for (name, sub, t) in binds:
                self.init_var(name, sub, t)
            return
        if self.needs_copy(s.init, nt):
            val = "_cp(%s)" % val
        st = self.strip(nt)
        if self.c.checks and (s.ty is not None or self.c.strict) and isinstance(st, str) and st in INT and s.init.k != "int":
            val = "_chk(%s, %r)" % (val, st)
        self.bind(s.pat, val, nt)

    def _declare_pat_names(self, pat, ty):
        if pat.k in ("pident", "pbind"):
            ent = self.declare(pat.name, ty)
            if ent["boxed"]:
                self.emit("%s = _Cell(None)" % ent["py"])
        elif pat.k == "ptuple":
            for p in pat.els:
                self._declare_pat_names(p, None)

    def needs_copy(self, e, t):
        """by-value use of a place whose value may be an array / Copy struct"""
        while e.k == "paren":
            e = e.e
        if e.k not in ("pathx", "field", "index", "deref"):
            return False
        if self.is_scalar_ty(t):
            return False
        if isinstance(t, tuple) and t[0] in ("ref", "ptr", "tup", "opt"):
            return False
        if e.k == "index" and e.i.k == "range":
            return False
        if e.k == "pathx" and len(e.path.segs) == 1:
            ent = self.lookup(e.path.segs[0])
            if ent is not None and isinstance(ent["ty"], tuple) and ent["ty"][0] == "ref":
                return False
            if ent is None and e.path.segs[0] not in self.c.consts:
                return False
        st = self.strip(t)
        if isinstance(st, str) and st in self.c.structs and not self.c.structs[st].cls._is_copy:
            return False
        if isinstance(st, str) and (st in self.c.shims or st in self.c.enums):
            return False
        return True

    def simple(self, val):
        """make sure an expression string is a plain name (evaluate once)"""
        if val.replace("_", "a").replace(".", "a").isalnum() and not val[0].isdigit():
            return val
        t = self.tmp()
        self.emit("%s = %s" % (t, val))
        return t

    def expr_stmt(self, e):
        k = e.k
        if k == "if":
            self.if_stmt(e)
        elif k == "block":
            self.block_stmts(e)
        elif k == "unsafe":
            self.block_stmts(e.body)
        elif k == "match":
            self.match(e, want=False)
        elif k in ("for", "while", "loop"):
            self.loop(e, want=False)
        elif k == "assign":
            self.assign(e)
        elif k == "return":
            if e.e is None:
                self.emit("return None")
            else:
                self.emit("return " + self.ret_wrap(self.ex(e.e)))
        elif k == "break":
            self.brk(e)
        elif k == "continue":
            self.cont(e)
        elif k == "macro":
            v = self.macro(e, want=False)
            if v:
                self.emit(v)
        else:
            v = self.ex(e)
            if v and v != "None":
                self.emit(v)

    # ------------------------------------------------------------ control flow
    def cond_expr(self, cnode):
        """returns (cond string, binds) -- emits prelude if necessary"""
        if cnode.k == "iflet":
            v = self.simple(self.ex(cnode.e))
            return self.pat_cond(cnode.pat, v)
        return self.ex(cnode), []

    def if_stmt(self, e):
        cond, binds = self.cond_expr(e.cond)
        self.emit("if %s:" % cond)
        self.ind += 1
        self.push()
        for (name, sub, t) in binds:
            self.init_var(name, sub, t)
        self.block_stmts(e.then)
        self.pop()
        self.ind -= 1
        if e.els is not None:
            self.emit("else:")
            self.ind += 1
            if e.els.k == "if":
                self.if_stmt(e.els)
            else:
                self.block_stmts(e.els)
            self.ind -= 1

    def if_expr(self, e):
        # try the ternary form
        if e.els is not None and e.cond.k != "iflet":
            c_lines, c = self.capture(lambda: self.ex(e.cond))
            a_lines, a = self.capture(lambda: self.branch_value(e.then))
            b_lines, b = self.capture(lambda: self.branch_value(e.els))
            if not a_lines and not b_lines and a is not None and b is not None:
                self.lines.extend(c_lines)
                return "(%s if %s else %s)" % (a, c, b)
        r = self.tmp()
        self._if_into(e, r)
        return r

    def branch_value(self, b):
        if b.k == "if":
            return self.if_expr(b)
        return self.block_value(b)

    def _if_into(self, e, r):
        cond, binds = self.cond_expr(e.cond)
        self.emit("if %s:" % cond)
        self.ind += 1
        self.push()
        for (name, sub, t) in binds:
            self.init_var(name, sub, t)
        v = self.block_value(e.then)
        self.emit("%s = %s" % (r, v))
        self.pop()
        self.ind -= 1
        self.emit("else:")
        self.ind += 1
        if e.els is None:
            self.emit("%s = None" % r)
        elif e.els.k == "if":
            self._if_into(e.els, r)
        else:
            v = self.block_value(e.els)
            self.emit("%s = %s" % (r, v))
        self.ind -= 1

    def match(self, e, want=True):
        v = self.simple(self.ex(e.e))
        sty = self.ty(e.e)
        r = self.tmp() if want else None
        guards = any(g is not None for (_, g, _) in e.arms)
        flag = self.tmp("_m") if guards else None
        if guards:
            self.emit("%s = False" % flag)
        first = True
        for (pat, guard, body) in e.arms:
            self.expected_pat_ty = sty
            cond, binds = self.pat_cond(pat, v, sty)
            if guards:
                self.emit("if not %s and (%s):" % (flag, cond))
            else:
                if cond == "True" and not first:
                    self.emit("else:")
                else:
                    self.emit("%s %s:" % ("if" if first else "elif", cond))
            first = False
            self.ind += 1
            self.push()
            for (name, sub, t) in binds:
                self.init_var(name, sub, t)
            if guard is not None:
                g = self.ex(guard)
                self.emit("if %s:" % g)
                self.ind += 1
                self.emit("%s = True" % flag)
            elif guards:
                self.emit("%s = True" % flag)
            if want:
                if body.k == "block":
                    val = self.block_value(body)
                else:
                    val = self.ex(body) if body.k not in ("return", "break", "continue") else \
                        (self.expr_stmt(body) or "None")
                self.emit("%s = %s" % (r, val))
            else:
                n0 = len(self.lines)
                if body.k == "block":
                    self.block_stmts(body)
                else:
                    self.expr_stmt(body)
                if len(self.lines) == n0:
                    self.emit("pass")
            if guard is not None:
                self.ind -= 1
            self.pop()
            self.ind -= 1
        if not guards and not any(self._irrefutable(p) for (p, _, _) in e.arms):
            self.emit("else:")
            self.emit("    raise _Panic('no match arm for %%r' %% (%s,))" % v)
        return r

    def _irrefutable(self, p):
        return p.k in ("pwild",) or (p.k == "pident" and not self._is_const_pat(p.name)) or \
            (p.k == "ptuple" and all(self._irrefutable(x) for x in p.els))

    def loop(self, e, want=True):
        k = e.k
        label = getattr(e, "label", None)
        r = self.tmp() if (want and k == "loop") else None
        if r:
            self.emit("%s = None" % r)
        ctx = {"label": label, "result": r, "flags": {}}
        if k == "for":
            it = e.it
            while it.k == "paren":
                it = it.e
            if it.k == "range" and it.lo is not None and it.hi is not None:
                lo, hi = self.seq([it.lo, it.hi])
                src = "range(%s, %s)" % (lo, hi + (" + 1" if it.incl else ""))
            elif it.k == "ref" and it.mut:
                src = "_ii(%s, True)" % self.ex(it.e)
            else:
                itt = self.ty(it)
                # `for x in s` with s: &mut [T] iterates by mutable reference
                mutit = isinstance(itt, tuple) and itt[0] == "ref" and len(itt) > 2 and itt[2] and \
                    isinstance(itt[1], tuple) and itt[1][0] == "arr"
                src = "_ii(%s%s)" % (self.ex(it), ", True" if mutit else "")
            simple_pat = e.pat.k == "pident" and not self._is_const_pat(e.pat.name) and e.pat.name not in self.boxed
            self.push()
            if simple_pat:
                ent = self.declare(e.pat.name, "usize" if it.k == "range" else None)
                self.emit("for %s in %s:" % (ent["py"], src))
                self.ind += 1
            else:
                t = self.tmp("_it")
                self.emit("for %s in %s:" % (t, src))
                self.ind += 1
                self.bind(e.pat, t, None)
            self.loops.append(ctx)
            self.block_stmts(e.body)
            self.loops.pop()
            self.ind -= 1
            self.pop()
        elif k == "while":
            if e.let is not None:
                self.emit("while True:")
                self.ind += 1
                v = self.simple(self.ex(e.let[1]))
                cond, binds = self.pat_cond(e.let[0], v)
                self.emit("if not (%s):" % cond)
                self.emit("    break")
                self.push()
                for (name, sub, t) in binds:
                    self.init_var(name, sub, t)
                self.loops.append(ctx)
                self.block_stmts(e.body)
                self.loops.pop()
                self.pop()
                self.ind -= 1
            else:
                c_lines, c = self.capture(lambda: self.ex(e.cond))
                if c_lines:
                    self.emit("while True:")
                    self.ind += 1
                    # re-emit with the right indentation
                    c = self.ex(e.cond)
                    self.emit("if not (%s):" % c)
                    self.emit("    break")
                else:
                    self.emit("while %s:" % c)
                    self.ind += 1
                self.loops.append(ctx)
                self.block_stmts(e.body)
                self.loops.pop()
                self.ind -= 1
        else:
            self.emit("while True:")
            self.ind += 1
            self.loops.append(ctx)
            self.block_stmts(e.body)
            self.loops.pop()
            self.ind -= 1
        # labeled jumps that crossed this loop
        for lab, (bflag, cflag) in ctx["flags"].items():
            if lab == label:
                for fl in (bflag, cflag):
                    if fl:
                        self.emit("%s = False" % fl)
                continue
            if bflag:
                self.emit("if %s:" % bflag)
                self.emit("    break")
            if cflag:
                outer = [c for c in self.loops if c["label"] == lab][0]
                direct = self.loops and self.loops[-1] is outer
                self.emit("if %s:" % cflag)
                if direct:
                    self.emit("    %s = False" % cflag)
                    self.emit("    continue")
                else:
                    self.emit("    break")
            if self.loops:
                self.loops[-1]["flags"].setdefault(lab, (bflag, cflag))
        return r

    def _label_flags(self, label, kind):
        # make sure the flag pair exists on every loop between here and the labelled one
        target = None
        for cx in reversed(self.loops):
            if cx["label"] == label:
                target = cx
                break
        if target is None:
            self.err("unknown loop label %s" % label)
        cur = self.loops[-1]
        bf, cf = cur["flags"].get(label, (None, None))
        if kind == "break" and bf is None:
            bf = [c["flags"][label][0] for c in self.loops if label in c["flags"] and c["flags"][label][0]]
            bf = bf[0] if bf else self.tmp("_brk")
        if kind == "continue" and cf is None:
            cf = [c["flags"][label][1] for c in self.loops if label in c["flags"] and c["flags"][label][1]]
            cf = cf[0] if cf else self.tmp("_cnt")
        cur["flags"][label] = (bf, cf)
        return target, bf, cf

    def brk(self, e):
        if not self.loops:
            self.err("break outside loop")
        if e.label is not None and self.loops[-1]["label"] != e.label:
            target, bf, cf = self._label_flags(e.label, "break")
            if e.e is not None:
                self.emit("%s = %s" % (target["result"], self.ex(e.e)))
            self.flagnames.add(bf)
            self.emit("%s = True" % bf)
            self.emit("break")
            return
        cx = self.loops[-1]
        if e.e is not None:
            self.emit("%s = %s" % (cx["result"], self.ex(e.e)))
        self.emit("break")

    def cont(self, e):
        if e.label is not None and self.loops[-1]["label"] != e.label:
            target, bf, cf = self._label_flags(e.label, "continue")
            self.flagnames.add(cf)
            self.emit("%s = True" % cf)
            self.emit("break")
            return
        self.emit("continue")


    # ------------------------------------------------------------ patterns
    def _is_const_pat(self, name):
        if self.lookup_item(name) is not None:
            return True
        if name in self.c.consts or name in self.c.variants or name == "None":
            return True
        return False

    def bind(self, pat, val, ty=None):
        """irrefutable binding"""
        k = pat.k
        if k == "pwild":
            if not val.replace("_", "a").replace(".", "a").isalnum():
                self.emit(val)
            return
        if k == "pident" or (k == "pbind" and pat.sub is None):
            self.init_var(pat.name, val, ty)
            return
        if k == "pbind":
            v = self.simple(val)
            self.init_var(pat.name, v, ty)
            self.bind(pat.sub, v, ty)
            return
        if k == "pref":
            inner = ty[1] if isinstance(ty, tuple) and ty[0] == "ref" else ty
            self.bind(pat.inner, "_deref(%s)" % val, inner)
            return
        if k == "ptuple":
            v = self.simple(val)
            st = self.strip(ty)
            for i, p in enumerate(pat.els):
                et = st[1][i] if isinstance(st, tuple) and st[0] == "tup" and i < len(st[1]) else None
                self.bind(p, "%s[%d]" % (v, i), et)
            return
        if k == "ptstruct" or k == "pstruct" or k == "pslice":
            v = self.simple(val)
            cond, binds = self.pat_cond(pat, v, ty)
            for (name, sub, t) in binds:
                self.init_var(name, sub, t)
            return
        self.err("unsupported irrefutable pattern %s" % k)

    def pat_cond(self, pat, v, ty=None):
        """refutable pattern on the (simple) expression v: (cond, [(name, expr, type)])"""
        k = pat.k
        if k == "pwild" or k == "prest":
            return "True", []
        if k == "pident":
            if self._is_const_pat(pat.name):
                return "%s == %s" % (v, self.ex(N("pathx", path=N("path", segs=[pat.name], gen={})))), []
            return "True", [(pat.name, v, ty)]
        if k == "pbind":
            if pat.sub is None:
                return "True", [(pat.name, v, ty)]
            c, b = self.pat_cond(pat.sub, v, ty)
            return c, [(pat.name, v, ty)] + b
        if k == "plit":
            return "%s == %s" % (v, self.ex(pat.e)), []
        if k == "prange":
            lo = self.ex(pat.lo)
            if pat.hi is None:
                return "%s <= %s" % (lo, v), []
            hi = self.ex(pat.hi)
            return "%s <= %s %s %s" % (lo, v, "<=" if pat.incl else "<", hi), []
        if k == "pref":
            inner = ty[1] if isinstance(ty, tuple) and ty[0] == "ref" else ty
            if pat.inner.k in ("pident", "pbind", "pwild"):
                return self.pat_cond(pat.inner, "_deref(%s)" % v, inner)
            t = self.tmp()
            c, b = self.pat_cond(pat.inner, t, inner)
            return "((%s := _deref(%s)), %s)[1]" % (t, v, c), b
        if k == "ptuple":
            conds, binds = [], []
            st = self.strip(ty)
            for i, p in enumerate(pat.els):
                et = st[1][i] if isinstance(st, tuple) and st[0] == "tup" and i < len(st[1]) else None
                c, b = self.pat_cond(p, "%s[%d]" % (v, i), et)
                if c != "True":
                    conds.append(c)
                binds += b
            return " and ".join(conds) or "True", binds
        if k == "por":
            cs = []
            binds = []
            for a in pat.alts:
                c, b = self.pat_cond(a, v, ty)
                cs.append("(%s)" % c)
                binds = binds or b
            return " or ".join(cs), binds
        if k == "ppath":
            return "%s == %s" % (v, self.ex(N("pathx", path=pat.path))), []
        if k == "ptstruct":
            segs = pat.path.segs
            name = segs[-1]
            if name in ("Some", "Ok", "Err"):
                conds = ["%s.var == %r" % (v, name)]
                c, b = self.pat_cond(pat.els[0], "%s.p[0]" % v, None)
                if c != "True":
                    conds.append(c)
                return " and ".join(conds), b
            ename = self._enum_of(segs)
            if ename:
                conds = ["%s.var == %r" % (v, name)]
                binds = []
                for i, p in enumerate(pat.els):
                    if p.k == "prest":
                        break
                    c, b = self.pat_cond(p, "%s.p[%d]" % (v, i), None)
                    if c != "True":
                        conds.append(c)
                    binds += b
                return " and ".join(conds), binds
            # tuple struct
            sname = self.info.owner if name == "Self" else name
            conds, binds = [], []
            for i, p in enumerate(pat.els):
                ft = None
                if sname in self.c.structs:
                    ft = self.norm(self.c.structs[sname].ftypes.get(str(i)))
                c, b = self.pat_cond(p, "%s._%d" % (v, i), ft)
                if c != "True":
                    conds.append(c)
                binds += b
            return " and ".join(conds) or "True", binds
        if k == "pstruct":
            segs = pat.path.segs
            name = segs[-1]
            ename = self._enum_of(segs)
            if ename:
                _, d, kind, fields = self.c.enum_value(ename, name)
                order = [f for f, _ in fields]
                conds = ["%s.var == %r" % (v, name)]
                binds = []
                for (f, p) in pat.fields:
                    c, b = self.pat_cond(p, "%s.p[%d]" % (v, order.index(f)), None)
                    if c != "True":
                        conds.append(c)
                    binds += b
                return " and ".join(conds), binds
            sname = self.info.owner if name == "Self" else name
            conds, binds = [], []
            for (f, p) in pat.fields:
                ft = None
                if sname in self.c.structs:
                    ft = self.norm(self.c.structs[sname].ftypes.get(f))
                c, b = self.pat_cond(p, "%s.%s" % (v, pyfield(f)), ft)
                if c != "True":
                    conds.append(c)
                binds += b
            return " and ".join(conds) or "True", binds
        if k == "pslice":
            conds = ["len(%s) == %d" % (v, len(pat.els))] if not any(p.k == "prest" for p in pat.els) else []
            binds = []
            for i, p in enumerate(pat.els):
                if p.k == "prest":
                    self.err("slice rest pattern")
                c, b = self.pat_cond(p, "%s[%d]" % (v, i), None)
                if c != "True":
                    conds.append(c)
                binds += b
            return " and ".join(conds) or "True", binds
        self.err("unsupported pattern %s" % k)

    def _enum_of(self, segs):
        name = segs[-1]
        if len(segs) >= 2:
            e = segs[-2]
            if e == "Self":
                e = self.info.owner
            if e in self.c.enums:
                return e
        cands = self.c.variants.get(name)
        if cands and name not in self.c.structs:
            return cands[0]
        return None

    # ------------------------------------------------------------ sequencing
    def seq(self, nodes, fns=None):
        """compile expressions left to right, preserving evaluation order when later
        ones need statements"""
        res = []
        for i, n in enumerate(nodes):
            f = (fns[i] if fns else None) or (lambda n=n: self.ex(n))
            res.append(self.capture(f))
        last = -1
        for i, (ls, _) in enumerate(res):
            if ls:
                last = i
        out = []
        for i, (ls, e) in enumerate(res):
            self.lines.extend(ls)
            if i < last and not self._pure(e):
                t = self.tmp()
                self.emit("%s = %s" % (t, e))
                e = t
            out.append(e)
        return out

    @staticmethod
    def _pure(e):
        s = e.replace("_", "a").replace(".", "a")
        return s.isalnum() or (s.startswith("-") and s[1:].isalnum())

    # ------------------------------------------------------------ expressions
    expected = None

    def ex(self, e):
        k = e.k
        m = getattr(self, "x_" + k, None)
        if m is None:
            self.err("expression kind %s" % k)
        return m(e)

    def x_int(self, e):
        return str(e.v)

    def x_float(self, e):
        return repr(e.v)

    def x_bool(self, e):
        return "True" if e.v else "False"

    def x_str(self, e):
        return repr(e.v)

    def x_char(self, e):
        return repr(e.v)

    def x_paren(self, e):
        return "(%s)" % self.ex(e.e)

    def x_tuple(self, e):
        if not e.els:
            return "None"
        vals = self.seq(e.els)
        return "(%s,)" % ", ".join(vals)

    def x_array(self, e):
        exp = self.expected
        el_exp = exp[1] if isinstance(exp, tuple) and exp[0] == "arr" else None
        def mk(n):
            def f():
                self.expected = el_exp
                return self.ex(n)
            return f
        vals = self.seq(e.els, [mk(n) for n in e.els])
        self.expected = exp
        return "_array(%s)" % ", ".join(vals)

    def x_repeat(self, e):
        exp = self.expected
        self.expected = exp[1] if isinstance(exp, tuple) and exp[0] == "arr" else None
        v, n = self.seq([e.e, e.n])
        self.expected = exp
        return "_repeat(%s, %s)" % (v, n)

    def x_range(self, e):
        lo = self.ex(e.lo) if e.lo is not None else "None"
        hi = self.ex(e.hi) if e.hi is not None else "None"
        return "_Rg(%s, %s%s)" % (lo, hi, ", True" if e.incl else "")

    def x_block(self, e):
        v = self.block_value(e)
        return v if v is not None else "None"

    def x_unsafe(self, e):
        return self.x_block(e.body)

    def x_if(self, e):
        return self.if_expr(e)

    def x_match(self, e):
        return self.match(e, want=True)

    def x_loop(self, e):
        return self.loop(e, want=True) or "None"

    x_while = x_loop
    x_for = x_loop

    def x_return(self, e):
        self.expr_stmt(e)
        return "None"

    def x_break(self, e):
        self.brk(e)
        return "None"

    def x_continue(self, e):
        self.cont(e)
        return "None"

    def x_assign(self, e):
        self.assign(e)
        return "None"

    def x_try(self, e):
        v = self.simple(self.ex(e.e))
        self.emit("if %s.var in ('None', 'Err'):" % v)
        self.emit("    return %s" % v)
        return "%s.p[0]" % v

    def x_cast(self, e):
        t = e.ty
        src_t = self.strip(self.ty(e.e))
        v = self.ex(e.e)
        if t.k == "tptr":
            inner = t.inner
            if inner.k == "tarray":
                n = self.ex(inner.n)
                return "_raw(%s, %s)" % (v, n)
            return v
        if t.k == "tinfer" or t.k == "tfn":
            return v
        nt = self.norm(t)
        if isinstance(nt, str):
            if nt in self.type_params:
                return "_cast(%s, _g[%r])" % (v, nt)
            if nt in INT:
                if isinstance(src_t, str) and (src_t == nt or fits(src_t, nt)):
                    return v
                if e.e.k == "int" and R.int_min(nt) <= e.e.v <= R.int_max(nt):
                    return v
                return "_cast(%s, %r)" % (v, nt)
            if nt in ("f32", "f64", "bool", "char"):
                return "_cast(%s, %r)" % (v, nt)
        self.err("cast to %r" % (nt,))

    def x_unary(self, e):
        v = self.ex(e.e)
        if e.op == "-":
            return "(-%s)" % v
        t = self.strip(self.ty(e.e))
        if t == "bool":
            return "(not %s)" % v
        if isinstance(t, str) and t in INT and t[0] == "u":
            return "(%s ^ %d)" % (v, R.int_max(t))
        if isinstance(t, str) and t in INT:
            return "(~%s)" % v
        return "_not(%s)" % v

    def x_deref(self, e):
        inner = e.e
        v = self.ex(inner)
        t = self.ty(inner)
        # a by-value scalar that is known not to be a reference needs no call
        return "_deref(%s)" % v

    def x_ref(self, e):
        inner = e.e
        while inner.k == "paren":
            inner = inner.e
        if not e.mut:
            return self.ex(inner)
        if inner.k == "pathx" and len(inner.path.segs) == 1:
            ent = self.lookup(inner.path.segs[0])
            if ent is not None and ent["boxed"]:
                if isinstance(ent["ty"], tuple) and ent["ty"][0] == "ref":
                    return "%s.v" % ent["py"]
                return "_refl(%s)" % ent["py"]
            return self.ex(inner)
        if inner.k == "index" and inner.i.k != "range":
            c, i = self.seq([inner.e, inner.i])
            return "_refi(%s, %s)" % (c, i)
        if inner.k == "field":
            o = self.ex(inner.e)
            t = self.strip(self.ty(inner.e))
            if inner.name.isdigit() and not (isinstance(t, str) and t in self.c.structs):
                return "_tf(%s, %s)" % (o, inner.name)
            return "_reff(%s, %r)" % (o, pyfield(inner.name))
        if inner.k == "deref":
            return self.ex(inner.e)
        return self.ex(inner)

    def x_bin(self, e):
        op = e.op
        if op in ("&&", "||"):
            l = self.ex(e.l)
            r_lines, r = self.capture(lambda: self.ex(e.r))
            if not r_lines:
                return "(%s %s %s)" % (l, "and" if op == "&&" else "or", r)
            t = self.tmp()
            self.emit("%s = %s" % (t, l))
            self.emit("if %s%s:" % ("" if op == "&&" else "not ", t))
            self.ind += 1
            r = self.ex(e.r)
            self.emit("%s = %s" % (t, r))
            self.ind -= 1
            return t
        l, r = self.seq([e.l, e.r])
        if op == "/" or op == "%":
            lt, rt = self.strip(self.ty(e.l)), self.strip(self.ty(e.r))
            t = lt if lt is not None else rt
            if isinstance(t, str) and t in INT and t[0] == "u":
                return "(%s %s %s)" % (l, "//" if op == "/" else "%", r)
            if t in ("f32", "f64") and op == "/":
                return "(%s / %s)" % (l, r)
            return "%s(%s, %s)" % ("_div" if op == "/" else "_rem", l, r)
        if self.c.strict and self.c.checks and op in ("+", "-", "*"):
            # strict: every arithmetic result of a statically known integer type is range-checked,
            # as a debug build of the reference would
            lt, rt = self.strip(self.ty(e.l)), self.strip(self.ty(e.r))
            t = lt if lt is not None else rt
            if isinstance(t, str) and t in INT:
                return "_chk((%s %s %s), %r)" % (l, op, r, t)
        return "(%s %s %s)" % (l, op, r)

    def x_index(self, e):
        c, i = self.seq([e.e, e.i])
        return "%s[%s]" % (c, i)

    def x_field(self, e):
        o = self.ex(e.e)
        if e.name.isdigit():
            t = self.strip(self.ty(e.e))
            if isinstance(t, tuple) and t[0] == "tup":
                return "%s[%s]" % (o, e.name)
            if isinstance(t, str) and t in self.c.structs:
                return "%s._%s" % (o, e.name)
            return "_tf(%s, %s)" % (o, e.name)
        return "%s.%s" % (o, pyfield(e.name))

    def x_structlit(self, e):
        segs = e.path.segs
        name = segs[-1]
        if name == "Self":
            name = self.info.owner
        ename = self._enum_of(segs) if name not in self.c.structs else None
        if ename:
            _, d, kind, fields = self.c.enum_value(ename, name)
            order = [f for f, _ in fields]
            byname = dict(e.fields)
            vals = self.seq([byname[f] for f in order])
            return "_E(%r, %r, %d, (%s,))" % (ename, name, d, ", ".join(vals))
        st = self.c.structs.get(name)
        shim = self.c.shims.get(name)
        if len(segs) >= 2 and segs[-2][:1].isupper() and segs[-2] != "Self":
            if self.c.autoload(segs[-2]):
                return self.x_structlit(e)
            self.err("unknown enum %s" % segs[-2])
        if st is None and shim is None:
            if self.c.autoload(name):
                return self.x_structlit(e)
            self.err("unknown struct %s" % name)
        def mk(f, n):
            def fn():
                self.expected = self.norm(st.ftypes.get(f)) if st else None
                v = self.ex(n)
                if self.needs_copy(n, self.expected if self.expected is not None else self.ty(n)):
                    v = "_cp(%s)" % v
                return v
            return fn
        save = self.expected
        vals = self.seq([n for _, n in e.fields], [mk(f, n) for f, n in e.fields])
        self.expected = save
        kw = ", ".join("%s=%s" % (pyfield(f), v) for (f, _), v in zip(e.fields, vals))
        cls = ("S_" + name) if st else "R.%s" % shim.__name__
        if e.base is not None:
            self.expected = name
            b = self.ex(e.base)
            self.expected = save
            return "_with(%s, %s)" % (b, kw)
        return "%s(%s)" % (cls, kw)

    def x_closure(self, e):
        name = self.tmp("_cl")
        params = []
        binds = []
        self.push()
        self.frames.append(Frame())
        hints = getattr(self, "closure_param_hint", None)
        self.closure_param_hint = None
        for i, (pat, ty) in enumerate(e.params):
            nt = self.norm(ty) if ty is not None else None
            if nt is None and hints and i < len(hints):
                nt = hints[i]      # the callee's signature fixes it (Aligned::from_fn: FnMut(usize) -> T)
            is_ref = isinstance(nt, tuple) and nt[0] == "ref"
            if pat.k == "pident" and (pat.name not in self.boxed or is_ref) and not self._is_const_pat(pat.name):
                ent = self.declare(pat.name, nt)
                if is_ref:
                    # a reference parameter that shares its name with a boxed local of the enclosing
                    # function (`best: &mut T` beside `let mut best`): the closure's name is the reference
                    ent["boxed"] = False
                params.append(ent["py"])
            else:
                a = "_p%d_%d" % (i, self.tmpn)
                params.append(a)
                binds.append((pat, a, nt))
        save_lines, save_ind, save_loops = self.lines, self.ind, self.loops
        self.lines, self.ind, self.loops = [], save_ind + 1, []
        for (pat, a, nt) in binds:
            self.bind(pat, a, nt)
        save_ret = self.ret_ty
        self.ret_ty = None
        if e.body.k == "block":
            v = self.block_value(e.body)
        else:
            v = self.ex(e.body)
        if v is not None:
            self.emit("return " + v)
        self.ret_ty = save_ret
        body = self.lines
        fr = self.frames.pop()
        self.lines, self.ind, self.loops = save_lines, save_ind, save_loops
        self.pop()
        self.emit("def %s(%s):" % (name, ", ".join(params)))
        outer_names = sorted(n for n in fr.assigned_outer)
        if outer_names:
            self.emit("    nonlocal " + ", ".join(outer_names))
            for n in outer_names:
                if n not in self.frames[-1].declared:
                    self.frames[-1].assigned_outer.add(n)
        if not body:
            self.emit("    pass")
        self.lines.extend(body)
        return name

    # ------------------------------------------------------------ paths
    def x_pathx(self, e):
        p = e.path
        segs = p.segs
        if len(segs) == 1:
            name = segs[0]
            ent = self.lookup(name)
            if ent is not None:
                return ent["py"] + (".v" if ent["boxed"] else "")
            if name in self.const_params:
                return "_g[%r]" % name
            li = self.lookup_item(name)
            if li is not None:
                if li[0] == "const":
                    return li[1]
                return self.fn_value(li[1], p)
            if name == "None":
                return "_NONE"
            if name in self.c.consts:
                return self.c.const(name, self.info)
            info = self.c.find_fn(name, self.info)
            if info is not None:
                return self.fn_value(info, p)
            if name in self.c.variants:
                en = self.c.variants[name][0]
                pn, d, kind, fields = self.c.enum_value(en, name)
                if kind == "unit":
                    return pn
                return "(lambda *a: _E(%r, %r, %d, a))" % (en, name, d)
            if name in self.c.structs:
                return "S_" + name
            if name in ("Some", "Ok", "Err"):
                return "_" + name
            if name == "PhantomData":
                return "None"
            if self.c.autoload(name):
                return self.x_pathx(e)
            self.err("unresolved name %s" % name)
        return self.path_value(p)

    def fn_value(self, info, p):
        self.c.pyfn(info)
        g = self.gen_dict(info, p)
        return "_fv(%s, %s)" % (info.pyname, g)

    def gen_dict(self, info, p):
        """generic bindings passed to a callee"""
        ga = p.gen.get(len(p.segs) - 1) if p is not None else None
        own = [g for g in info.node.gens]
        if not ga:
            # No turbofish: the callee's parameters are inferred from the arguments, which this
            # transpiler does not do -- the caller's bindings are handed on by NAME.  One case where
            # the same name means two things is all over the reference: a function generic over
            # `T: Pixel` calling one generic over `T: Coefficient` with T::Coeff buffers
            # (encode_tx_block -> forward_transform / quantize / dequantize / inverse_transform_add).
            mine = {g[1]: g[2] for g in self.info.gens if g[0] == "type" and g[2]}
            for g in own:
                if g[0] == "type" and g[2] and "Coefficient" in g[2] and "Pixel" in mine.get(g[1], ()):
                    return "{**_g, %r: ('i16' if _g[%r] == 'u8' else 'i32')}" % (g[1], g[1])
            return "_g"
        items = []
        for g, a in zip(own, ga):
            if a.k == "gconst":
                items.append("%r: %s" % (g[1], self.ex(a.e)))
            else:
                nt = self.norm(a)
                if g[0] == "const":
                    # a const generic argument written as a bare path
                    items.append("%r: %s" % (g[1], self.ex(N("pathx", path=a))))
                elif isinstance(nt, str) and nt in self.type_params:
                    items.append("%r: _g[%r]" % (g[1], nt))
                else:
                    items.append("%r: %r" % (g[1], nt))
        return "{**_g, %s}" % ", ".join(items)

    def resolve_fn(self, p):
        segs = p.segs
        if len(segs) == 1:
            li = self.lookup_item(segs[0])
            if li is not None and li[0] == "fn":
                return li[1]
            if self.lookup(segs[0]) is not None:
                return None
            return self.c.find_fn(segs[0], self.info)
        a, b = segs[-2], segs[-1]
        if a == "Self":
            a = self.info.owner
        if a in self.c.methods or a in self.c.trait_impls:
            m = self.c.find_method(a, b)
            if m is not None:
                return m
        if a in self.c.structs or a in self.c.enums or a in PRIMS or a in self.type_params:
            return None
        # module-qualified free function
        c = self.c.fns.get(b)
        if c:
            for i in c:
                if a in i.modpath or os.path.basename(i.fname)[:-3] == a or \
                        os.path.basename(os.path.dirname(i.fname)) == a:
                    return i
            if a in ("self", "super", "crate", "rust"):
                return self.c.find_fn(b, self.info)
        return None

    def path_value(self, p):
        segs = p.segs
        a, b = segs[-2], segs[-1]
        if a == "Self":
            a = self.info.owner
        if a in INT:
            if b == "MAX":
                return str(R.int_max(a))
            if b == "MIN":
                return str(R.int_min(a))
            if b == "BITS":
                return str(INT[a])
        if a in ("f64", "f32"):
            import sys
            if b == "MAX":
                return repr(sys.float_info.max if a == "f64" else 3.4028234663852886e+38)
            if b == "EPSILON":
                return repr(sys.float_info.epsilon if a == "f64" else 1.1920929e-07)
            if b == "INFINITY":
                return "float('inf')"
        if (a in INT or a in ("f64", "f32")) and b in ("from", "cast_from"):
            return "(lambda _x: _cast(_x, %r))" % a
        if a in self.type_params and b in ("from", "cast_from"):
            return "(lambda _x: _cast(_x, _g[%r]))" % a
        if a in self.type_params:
            self.err("associated item %s::%s of a type parameter" % (a, b))
        key = a + "::" + b
        if key in self.c.consts:
            return self.c.const(key, self.info)
        if a in self.c.enums:
            pn, d, kind, fields = self.c.enum_value(a, b)
            if kind == "unit":
                return pn
            return "(lambda *a: _E(%r, %r, %d, a))" % (a, b, d)
        if a in self.c.aliases:
            na = self.norm(self.c.aliases[a])
            if isinstance(na, str):
                return self.path_value(N("path", segs=segs[:-2] + [na, b], gen=p.gen))
        info = self.resolve_fn(p)
        if info is not None:
            return self.fn_value(info, p)
        if b in self.c.consts and len(self.c.consts[b]) >= 1:
            return self.c.const(b, self.info)
        if a == "Ordering":
            return "_E('Ordering', %r, %d)" % (b, {"Less": -1, "Equal": 0, "Greater": 1}[b])
        if self.c.autoload(a) or self.c.autoload(b):
            return self.path_value(p)
        self.err("unresolved path %s" % "::".join(segs))

    # ------------------------------------------------------------ calls
    def args(self, nodes, info=None):
        fns = []
        for i, n in enumerate(nodes):
            def f(n=n, i=i):
                save = self.expected
                pty = None
                pmut = False
                if info is not None and i < len(info.node.params):
                    ppat, pt = info.node.params[i]
                    pty = FnCompiler(self.c, info).norm(pt)
                    # a by-value array / Copy-struct argument is copied only when the callee
                    # can mutate its parameter (`mut x: [T; N]`) or destructures it
                    pmut = not (ppat.k == "pident" or (ppat.k == "pbind" and not getattr(ppat, "mut", False)))
                    if isinstance(pty, tuple) and pty[0] in ("ref", "ptr"):
                        pmut = False
                self.expected = pty
                if n.k == "ref" and n.mut and isinstance(pty, tuple) and pty[0] == "ref" and \
                        self.is_scalar_ty(pty[1]) is False and False:
                    pass
                v = self.ex(n)
                self.expected = save
                if pmut and self.needs_copy(n, pty if pty is not None else self.ty(n)) and \
                        not self.is_scalar_ty(pty):
                    st = self.strip(pty)
                    if st is None or (isinstance(st, tuple) and st[0] == "arr") or \
                            (isinstance(st, str) and st in self.c.structs):
                        v = "_cp(%s)" % v
                return v
            fns.append(f)
        return self.seq(nodes, fns)

    def x_call(self, e):
        f = e.f
        if f.k == "pathx":
            return self.call_path(f.path, e.args)
        if f.k == "qpath":
            # <T as Trait>::f(args)
            t = self.norm(f.ty)
            return self.call_path(N("path", segs=[t if isinstance(t, str) else "?"] + f.rest, gen={}), e.args)
        fv = self.ex(f)
        vals = self.args(e.args)
        return "%s(%s)" % (fv, ", ".join(vals))

    def call_path(self, p, argn):
        segs = p.segs
        name = segs[-1]
        if len(segs) == 1:
            ent = self.lookup(name)
            if ent is not None:
                vals = self.args(argn)
                return "%s(%s)" % (ent["py"] + (".v" if ent["boxed"] else ""), ", ".join(vals))
            li = self.lookup_item(name)
            if li is not None and li[0] == "fn":
                return self.call_info(li[1], p, argn)
        b = self.builtin_call(p, argn)
        if b is not None:
            return b
        info = self.resolve_fn(p)
        if info is not None:
            return self.call_info(info, p, argn)
        # constructors
        if name == "Self" or (len(segs) == 1 and name in self.c.structs):
            sname = self.info.owner if name == "Self" else name
            st = self.c.structs[sname]
            def mk(i, n):
                def fn():
                    save = self.expected
                    self.expected = self.norm(st.ftypes.get(str(i)))
                    v = self.ex(n)
                    self.expected = save
                    return v
                return fn
            vals = self.seq(argn, [mk(i, n) for i, n in enumerate(argn)])
            return "S_%s(%s)" % (sname, ", ".join(vals))
        en = self._enum_of(segs)
        if en:
            pn, d, kind, fields = self.c.enum_value(en, name)
            vals = self.args(argn)
            return "_E(%r, %r, %d, (%s,))" % (en, name, d, ", ".join(vals))
        a = segs[-2] if len(segs) > 1 else None
        if a in self.c.shims:
            vals = self.args(argn)
            return "R.%s.%s(%s)" % (self.c.shims[a].__name__, name, ", ".join(vals))
        if self.c.autoload(name) or (a is not None and self.c.autoload(a)):
            return self.call_path(p, argn)
        if name == "default" and a in self.c.structs and not argn:   # #[derive(Default)]
            return self.default_for(a)
        self.err("unresolved call %s" % "::".join(segs))

    def call_info(self, info, p, argn):
        self.c.pyfn(info)
        vals = self.args(argn, info if not info.node.has_self else None)
        g = self.gen_dict(info, p)
        # impl-level generics given on the type segment: Type::<T>::f
        return "%s(%s)" % (info.pyname, ", ".join([g] + vals))

    def builtin_call(self, p, argn):
        segs = p.segs
        name = segs[-1]
        a = segs[-2] if len(segs) > 1 else None
        full = "::".join(segs)
        if a == "Self":
            a = self.info.owner
        if a == "CpuFeatureLevel" and name == "default" and not argn:
            return "None"   # the CPU dispatch level: carried around, never inspected by rust:: code
        if len(segs) == 3 and segs[0] in self.type_params and segs[1] == "Coeff" and name in ("cast_from", "from"):
            # <T as Pixel>::Coeff: i16 for u8 pixels, i32 for u16 (v_frame 0.3.9 pixel.rs)
            v = self.args(argn)[0]
            return "_cast(%s, 'i16' if _g[%r] == 'u8' else 'i32')" % (v, segs[0])
        if a is not None and a in self.c.aliases and a not in self.c.structs:
            na = self.norm(self.c.aliases[a])
            if isinstance(na, str):
                a = na
        gen0 = p.gen.get(len(segs) - 2) if len(segs) > 1 else None
        if name in ("cast_from", "from", "new", "into") and (a in INT or a in self.type_params or a in
                                                             ("f64", "f32", "Into", "From") or a in NONZERO):
            if a in ("Into", "From"):
                tgt = self.norm(gen0[0]) if gen0 else None
            elif a in NONZERO:
                v = self.args(argn)[0]
                if name == "new":
                    t = self.tmp()
                    self.emit("%s = %s" % (t, v))
                    return "(_Some(%s) if %s != 0 else _NONE)" % (t, t)
                return v
            else:
                tgt = a
            if name == "new" and tgt not in INT:
                return None
            v = self.args(argn)[0]
            if tgt is None:
                return v
            src_t = self.strip(self.ty(argn[0]))
            if tgt in self.type_params:
                return "_cast(%s, _g[%r])" % (v, tgt)
            if isinstance(src_t, str) and (src_t == tgt or fits(src_t, tgt)):
                return v
            return "_cast(%s, %r)" % (v, tgt)
        if name == "abs" and a in INT and len(argn) == 1:     # i16::abs(x), the UFCS form of x.abs()
            return "_chk(abs(%s), %r)" % (self.args(argn)[0], a)
        if name == "try_from" and (a in INT):
            return "_try_from(%s, %r)" % (self.args(argn)[0], a)
        if name in ("zero", "one", "max_value", "min_value", "default") and (a in INT or a in self.type_params):
            if name in ("zero", "default"):
                return "0"
            if name == "one":
                return "1"
            if a in INT:
                return str(R.int_max(a) if name == "max_value" else R.int_min(a))
            return "R.int_m%s(_g[%r])" % ("ax" if name == "max_value" else "in", a)
        if full in ("cmp::max", "max", "std::cmp::max", "core::cmp::max") and len(argn) == 2:
            x, y = self.args(argn)
            return "_im(%s, None, 'max', (%s,))" % (x, y)
        if full in ("cmp::min", "min", "std::cmp::min", "core::cmp::min") and len(argn) == 2:
            x, y = self.args(argn)
            return "_im(%s, None, 'min', (%s,))" % (x, y)
        if name == "msb" and len(argn) == 1 and not self.c.fns.get("msb"):
            return "_msb(%s)" % self.args(argn)[0]
        if name == "round_shift" and len(argn) == 2 and not self.c.fns.get("round_shift"):
            return "_round_shift(%s)" % ", ".join(self.args(argn))
        if name == "clamp" and len(argn) == 3 and not self.c.fns.get("clamp"):
            return "_clamp(%s)" % ", ".join(self.args(argn))
        if name in ("Some", "Ok", "Err") and len(segs) == 1:
            return "_%s(%s)" % (name, self.args(argn)[0])
        if full in ("mem::size_of_val", "std::mem::size_of_val", "size_of_val"):
            n = argn[0]
            while n.k in ("ref", "paren"):
                n = n.e
            t = self.strip(self.ty(n))
            if isinstance(t, str) and t in INT:
                return str(INT[t] // 8)
            self.err("size_of_val of unknown type")
        if name == "size_of" and a in ("mem", None):
            ga = p.gen.get(len(segs) - 1)
            t = self.norm(ga[0])
            if t in self.type_params:
                return "(R.INT_BITS[_g[%r]] // 8)" % t
            return str(INT[t] // 8)
        if full in ("mem::swap", "std::mem::swap"):
            x, y = self.args(argn)
            return "_swap(%s, %s)" % (x, y)
        if full in ("slice::from_raw_parts", "slice::from_raw_parts_mut", "std::slice::from_raw_parts",
                    "from_raw_parts", "from_raw_parts_mut"):
            return "_raw(%s)" % ", ".join(self.args(argn))
        if a in VECS and name in ("new", "with_capacity", "new_const"):
            self.args(argn)
            return "_S([])"
        if a in VECS and name == "from":
            return "%s.to_vec()" % self.args(argn)[0]
        if a in WRAPPERS and name in ("new", "from", "uninit", "uninit_array", "uninitialized", "zeroed"):
            if name in ("new", "from"):
                v = self.args(argn)[0]
                return "R.Aligned(%s)" % v if a == "Aligned" else v
            if gen0:
                t = self.norm(gen0[0]) if gen0[0].k != "gconst" else None
                if self.is_scalar_ty(t) or t in self.type_params:
                    return "0"
            exp = self.strip(self.expected)
            if exp is None and gen0 and a == "Aligned" and name == "uninit_array":
                # Aligned::<[MaybeUninit<X>; N]>::uninit_array() bound to an untyped `let`
                return "R.Aligned(%s)" % self.default_for(self.strip(self.norm(gen0[0])), uninit=True)
            if exp is None:
                if a == "Aligned" and name == "uninit_array":
                    # `let mut edge_buf = Aligned::uninit_array();` -- the type comes from the later
                    # call (inference this transpiler does not do).  Every such site of the reference
                    # is an IntraEdgeBuffer = Aligned<[MaybeUninit<T>; 4 * MAX_TX_SIZE + 1]>
                    # (src/partition.rs:600; api/lookahead.rs:57, encoder.rs:1476, rdo.rs:1437,1624)
                    return "R.Aligned(_S([0] * 257))"
                return "None"
            d = self.default_for(exp, uninit=True)
            return "R.Aligned(%s)" % d if (a == "Aligned" and name == "uninitialized") else d
        if name == "default" and (a == "Default" or a is None):
            return self.default_for(self.strip(self.expected))
        if a == "Aligned" and name == "from_fn":
            self.closure_param_hint = ["usize"]
            f = self.args(argn)[0]
            self.closure_param_hint = None
            exp = self.strip(self.expected)
            if exp is None:
                # `let edge_buf = Aligned::from_fn(|i| ..)` handed to IntraEdge::mock: an IntraEdgeMock
                # = Aligned<[T; 4 * MAX_TX_SIZE + 1]> (partition.rs:603; the untyped sites of the tree)
                return "R.Aligned(_S([(%s)(_i) for _i in range(257)]))" % f
            self.err("Aligned::from_fn")
        if full == "ILog::ilog" and len(argn) == 1:
            return "_im(%s, None, 'ilog', ())" % self.args(argn)[0]
        if name == "transmute" and len(argn) == 1:
            return self.args(argn)[0]
        if name == "new_unchecked" and a in NONZERO:
            return self.args(argn)[0]
        if name == "drop" and len(segs) == 1:
            self.args(argn)
            return "None"
        if name == "black_box":
            return self.args(argn)[0]
        if a == "f64" and name == "from":
            return "float(%s)" % self.args(argn)[0]
        return None

    def default_for(self, t, uninit=False):
        t = self.strip(t)
        if t is None:
            self.err("Default::default() with unknown expected type")
        if isinstance(t, str):
            if t in INT:
                return "0"
            if t in ("f32", "f64"):
                return "0.0"
            if t == "bool":
                return "False"
            if t in self.type_params or t == "Coeff":    # <T as Pixel>::Coeff: i16 / i32
                return "0"
            m = self.c.find_method(t, "default")
            if m is not None:
                self.c.pyfn(m)
                return "%s(_g)" % m.pyname
            if t in self.c.structs:
                st = self.c.structs[t]
                return "S_%s(%s)" % (t, ", ".join("%s=%s" % (pyfield(f), self.default_for(self.norm(ft)))
                                                  for f, ft in st.node.fields))
            if t in self.c.enums and getattr(self.c.enums[t], "default_variant", None):
                dv = self.c.enums[t].default_variant       # #[derive(Default)] + #[default]
                pn, d, kind, fields = self.c.enum_value(t, dv)
                if kind == "unit":
                    return pn
        if isinstance(t, tuple):
            if t[0] == "tup":
                return "(%s,)" % ", ".join(self.default_for(x) for x in t[1])
            if t[0] == "opt":
                return "_NONE"
            if t[0] == "arr":
                if len(t) > 2 and t[2] is not None:
                    return "_vrep(%s, %s)" % (self.default_for(t[1]), self.ex(t[2]))
                self.err("default for array of unknown length")
        self.err("default for %r" % (t,))

    # ------------------------------------------------------------ method calls
    ITER_MUT_ARGS = ("zip", "chain")

    def x_mcall(self, e):
        name = e.name
        recv_t = self.strip(self.ty(e.recv))
        # arguments of the form `&mut place` given to zip() iterate mutably
        if name == "zip" and len(e.args) == 1 and e.args[0].k == "ref" and e.args[0].mut:
            r, a = self.seq([e.recv, e.args[0].e])
            return "_mc(_g, %s, None, 'zip', (%s, True))" % (r, a)
        if name in ("sum", "product", "collect", "into", "try_into", "unwrap", "cast") and not e.args:
            pass
        rn = e.recv
        while rn.k == "paren":
            rn = rn.e
        if name == "write" and len(e.args) == 1 and rn.k == "index" and rn.i.k != "range":
            # MaybeUninit<T>::write on an element: store, and hand back a reference to it
            cc, ii, vv = self.seq([rn.e, rn.i, e.args[0]])
            return "R.mwrite(%s, %s, %s)" % (cc, ii, vv)
        recv = self.ex(e.recv) if e.recv.k != "ref" else self.ex(e.recv)
        hint = self.tyname(recv_t) if isinstance(recv_t, str) else None
        if name == "write" and len(e.args) == 1:
            # MaybeUninit<T>::write through an element reference
            pass
        vals = self.seq([N("_raw", s=recv)] + list(e.args),
                        [lambda: recv] + [None] * len(e.args))
        recv = vals[0]
        args = vals[1:]
        if name == "sum" and e.gen:
            pass
        # static fast paths for integers
        if isinstance(recv_t, str) and recv_t in INT:
            if name in ("min", "max") and len(args) == 1:
                return "%s(%s, %s)" % ("_imin" if name == "min" else "_imax", recv, args[0]) \
                    if False else "_im(%s, %r, %r, (%s,))" % (recv, recv_t, name, args[0])
            if name not in self.c.methods.get(recv_t, {}):
                return "_im(%s, %r, %r, (%s))" % (recv, recv_t, name, "".join(a + ", " for a in args))
        return "_mc(_g, %s, %s, %r, (%s))" % (recv, hint or "None", name, "".join(a + ", " for a in args))

    def x__raw(self, e):
        return e.s

    # ------------------------------------------------------------ assignment
    def assign(self, e):
        op = e.op
        l = e.l
        while l.k == "paren":
            l = l.e
        if op == "=":
            if l.k == "tuple":
                v = self.simple(self.ex(e.r))
                for i, sub in enumerate(l.els):
                    self.assign(N("assign", op="=", l=sub, r=N("_raw", s="%s[%d]" % (v, i))))
                return
            if l.k == "pathx" and len(l.path.segs) == 1 and l.path.segs[0] == "_":
                self.emit(self.ex(e.r))
                return
            save = self.expected
            self.expected = self.ty(l)
            val = self.ex(e.r)
            if e.r.k != "_raw" and self.needs_copy(e.r, self.expected if self.expected is not None else self.ty(e.r)):
                val = "_cp(%s)" % val
            self.expected = save
            self.store_to(l, val)
            return
        bop = op[:-1]
        lt = self.strip(self.ty(l))
        rnode = N("bin", op=bop, l=N("_raw", s="{L}"), r=e.r)

        OPN = {"+": "add", "-": "sub", "*": "mul", "/": "div", "%": "rem", "<<": "shl", ">>": "shr",
               "&": "bitand", "|": "bitor", "^": "bitxor"}

        def combine(cur, rv):
            if not self.is_scalar_ty(lt) and not (isinstance(lt, str) and lt in self.type_params) and \
                    (lt is None and self.strip(self.ty(e.r)) is None or
                     (isinstance(lt, str) and lt in self.c.structs)):
                pyop = {"/": "_div(a, b)", "%": "_rem(a, b)"}.get(bop, "a %s b" % bop)
                return "R.augop(%s, %r, %s, lambda a, b: %s)" % (cur, OPN[bop], rv, pyop)
            if bop in ("/", "%"):
                if isinstance(lt, str) and lt in INT and lt[0] == "u":
                    s = "(%s %s %s)" % (cur, "//" if bop == "/" else "%", rv)
                elif lt in ("f32", "f64") and bop == "/":
                    s = "(%s / %s)" % (cur, rv)
                else:
                    s = "%s(%s, %s)" % ("_div" if bop == "/" else "_rem", cur, rv)
            else:
                s = "(%s %s %s)" % (cur, bop, rv)
            if self.c.checks and isinstance(lt, str) and lt in INT and bop in ("+", "-", "*", "<<"):
                s = "_chk(%s, %r)" % (s, lt)
            return s
        if l.k == "pathx" and len(l.path.segs) == 1:
            ent = self.lookup(l.path.segs[0])
            if ent is None:
                self.err("assignment to unknown %s" % l.path.segs[0])
            rv = self.ex(e.r)
            cur = ent["py"] + (".v" if ent["boxed"] else "")
            if isinstance(ent["ty"], tuple) and ent["ty"][0] == "ref" and not ent["boxed"]:
                pass
            self.assign_var(ent, combine(cur, rv))
            return
        if l.k == "index":
            c, i, rv = self.seq([l.e, l.i, e.r])
            c, i = self.simple(c), (i if self._pure(i) else self.simple(i))
            self.emit("%s[%s] = %s" % (c, i, combine("%s[%s]" % (c, i), rv)))
            return
        if l.k == "field":
            o, rv = self.seq([l.e, e.r])
            o = self.simple(o)
            t = self.strip(self.ty(l.e))
            if l.name.isdigit() and not (isinstance(t, str) and t in self.c.structs):
                fld = "_%s" % l.name
            else:
                fld = pyfield(l.name)
            self.emit("%s.%s = %s" % (o, fld, combine("%s.%s" % (o, fld), rv)))
            return
        if l.k == "deref":
            r, rv = self.seq([l.e, e.r])
            r = self.simple(r)
            self.emit("_store(%s, %s)" % (r, combine("_deref(%s)" % r, rv)))
            return
        self.err("compound assignment target %s" % l.k)

    def store_to(self, l, val):
        if l.k == "pathx" and len(l.path.segs) == 1:
            ent = self.lookup(l.path.segs[0])
            if ent is None:
                self.err("assignment to unknown %s" % l.path.segs[0])
            self.assign_var(ent, val)
            return
        if l.k == "index":
            v = self.simple(val) if not self._pure(val) else val
            c, i = self.seq([l.e, l.i])
            self.emit("%s[%s] = %s" % (c, i, v))
            return
        if l.k == "field":
            v = self.simple(val) if not self._pure(val) else val
            o = self.ex(l.e)
            t = self.strip(self.ty(l.e))
            if l.name.isdigit() and not (isinstance(t, str) and t in self.c.structs):
                self.emit("%s._%s = %s" % (o, l.name, v))
            else:
                self.emit("%s.%s = %s" % (o, pyfield(l.name), v))
            return
        if l.k == "deref":
            v = self.simple(val) if not self._pure(val) else val
            r = self.ex(l.e)
            self.emit("_store(%s, %s)" % (r, v))
            return
        self.err("assignment target %s" % l.k)

    # ------------------------------------------------------------ macros
    def x_macro(self, e):
        return self.macro(e, want=True) or "None"

    def macro(self, e, want):
        n = e.name
        if n in ("assert", "debug_assert"):
            c = self.ex(e.args[0])
            self.emit("if not (%s):" % c)
            self.emit("    raise _Panic(%r)" % ("assertion failed in %s" % self.info.node.name))
            return None
        if n in ("assert_eq", "debug_assert_eq", "assert_ne", "debug_assert_ne"):
            a, b = self.seq(e.args[:2])
            self.emit("if %s(%s == %s):" % ("not " if n.endswith("eq") else "", a, b))
            self.emit("    raise _Panic('%s failed in %s: %%r vs %%r' %% (%s, %s))" % (n, self.info.node.name, a, b))
            return None
        if n == "izip":
            parts = []
            nodes = []
            muts = []
            for a in e.args:
                if a.k == "ref" and a.mut:
                    nodes.append(a.e)
                    muts.append(True)
                else:
                    nodes.append(a)
                    muts.append(False)
            vals = self.seq(nodes)
            for v, m in zip(vals, muts):
                parts.append("_ii(%s%s)" % (v, ", True" if m else ""))
            return "_izip(%s)" % ", ".join(parts)
        if n == "vec":
            if e.rep is not None:
                v, cnt = self.seq([e.args[0], e.rep])
                return "_vrep(%s, %s)" % (v, cnt)
            vals = self.seq(e.args)
            return "_S([%s])" % ", ".join(vals)
        if n in ("panic", "unreachable", "unimplemented", "todo"):
            self.emit("raise _Panic(%r)" % ("%s! in %s" % (n, self.info.node.name)))
            return None
        if n in ("println", "eprintln", "print", "write", "writeln", "format"):
            return "None"
        if n == "matches":
            v = self.simple(self.ex(e.args[0]))
            cond, binds = self.pat_cond(e.pat, v, self.ty(e.args[0]))
            if e.guard is not None:
                if binds:
                    self.err("matches! guard with bindings")
                cond = "(%s) and (%s)" % (cond, self.ex(e.guard))
            return "(%s)" % cond
        if n in ("search_pattern", "search_pattern_subpel") and e.args is None:
            # the two macro_rules! of src/me.rs:928-941, expanded as their bodies say:
            #   (fa: [a0, a1, ..], fb: [b0, b1, ..]) => [MotionVector { fa: a_k << 3, fb: b_k << 3 }, ..]
            # (search_pattern_subpel: without the shifts); the expansion is re-parsed as Rust
            toks = e.raw[1:-1]
            groups, names, i = [], [], 0
            while i < len(toks):
                assert toks[i].k == "id" and toks[i + 1].v == ":" and toks[i + 2].v == "[", toks[i:i + 3]
                names.append(toks[i].v)
                i += 3
                cur, els, depth = [], [], 0
                while not (toks[i].v == "]" and depth == 0):
                    if toks[i].v in ("(", "["):
                        depth += 1
                    elif toks[i].v in (")", "]"):
                        depth -= 1
                    if toks[i].v == "," and depth == 0:
                        els.append(cur)
                        cur = []
                    else:
                        cur.append(toks[i])
                    i += 1
                if cur:
                    els.append(cur)
                groups.append(els)
                i += 1
                if i < len(toks) and toks[i].v == ",":
                    i += 1
            assert len(groups) == 2 and len(groups[0]) == len(groups[1])
            sh = " << 3" if n == "search_pattern" else ""
            txt = lambda ts: " ".join(str(t.v[0]) + (t.v[1] or "") if t.k == "int" else str(t.v) for t in ts)
            src = "[" + ", ".join("MotionVector { %s: (%s)%s, %s: (%s)%s }" % (names[0], txt(a), sh, names[1], txt(b), sh)
                                  for a, b in zip(*groups)) + "]"
            from .lexer import lex
            return self.ex(Parser(lex(src), "<%s!>" % n).parse_expr())
        self.err("macro %s!" % n)
