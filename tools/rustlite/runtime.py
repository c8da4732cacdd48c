"""Run-time support for transpiled reference code (test infrastructure).

Value model
  integers   Python ints (casts wrap like Rust's `as`; the transpiler range-
             checks every value bound to a typed `let`, parameter or return,
             which is where a debug build of the reference would panic)
  bool/float Python bool / float (f32 values are rounded through struct.pack)
  [T; N], &[T], Vec<T>   RSlice: (backing list, offset, length) -- sub-slices
             alias their parent exactly as Rust borrows do
  *const T   RPtr: (backing list, index)
  &mut <scalar place>   Cell (boxed local) or RRef(container, key)
  iterators  RIter around a Python iterator
  Option / Result / user enums   REnum
  structs    instances of classes generated per `struct` (RStruct)
"""
import struct as _struct

INT_BITS = {"u8": 8, "u16": 16, "u32": 32, "u64": 64, "u128": 128, "usize": 64,
            "i8": 8, "i16": 16, "i32": 32, "i64": 64, "i128": 128, "isize": 64}
INT_TYPES = tuple(INT_BITS)


class Panic(Exception):
    """A Rust panic (failed assert, out-of-range index, overflow check)."""


def int_min(t):
    return -(1 << (INT_BITS[t] - 1)) if t[0] == "i" else 0


def int_max(t):
    return (1 << (INT_BITS[t] - 1)) - 1 if t[0] == "i" else (1 << INT_BITS[t]) - 1


def wrap(v, t):
    b = INT_BITS[t]
    v &= (1 << b) - 1
    if t[0] == "i" and v >> (b - 1):
        v -= 1 << b
    return v


def f32(x):
    return _struct.unpack("f", _struct.pack("f", x))[0]


class F32(float):
    """A Rust f32 value: every arithmetic operation it takes part in is rounded to f32 (IEEE
    single precision, one operation at a time -- Rust never contracts or re-associates).  The
    operation is carried out in f64 and rounded once more, which for + - * / of two f32 operands
    is the correctly rounded f32 result (53 >= 2 * 24 + 2 bits).  Untyped float literals are plain
    Python floats and adapt to the F32 operand, as Rust's literal inference does; an f64 never
    meets an f32 without an `as` cast in Rust source."""
    __slots__ = ()

    def __new__(cls, v=0.0):
        return float.__new__(cls, f32(float(v)))

    def __add__(self, o): return F32(float(self) + float(o))
    def __radd__(self, o): return F32(float(o) + float(self))
    def __sub__(self, o): return F32(float(self) - float(o))
    def __rsub__(self, o): return F32(float(o) - float(self))
    def __mul__(self, o): return F32(float(self) * float(o))
    def __rmul__(self, o): return F32(float(o) * float(self))
    def __truediv__(self, o): return F32(float(self) / float(o))
    def __rtruediv__(self, o): return F32(float(o) / float(self))
    def __neg__(self): return F32(-float(self))
    def __abs__(self): return F32(abs(float(self)))
    def __repr__(self): return "F32(%r)" % float(self)


def cast(v, t):
    """Rust `v as t`."""
    if t in INT_BITS:
        if isinstance(v, float):
            if v != v:
                return 0
            v = int(v)  # truncation toward zero, then saturate (Rust semantics)
            return max(int_min(t), min(int_max(t), v))
        if isinstance(v, REnum):
            v = v.disc
        elif isinstance(v, RStruct):
            raise Panic("cast of struct")
        return wrap(int(v), t)
    if t == "f64":
        return float(v)
    if t == "f32":
        return F32(float(v))
    if t == "bool":
        return bool(v)
    if t == "char":
        return v
    raise Panic("cast to %r" % (t,))


def chk(v, t, what=""):
    """Overflow check at a typed binding (a debug build would have panicked earlier)."""
    if t in INT_BITS and type(v) is int:
        if not (int_min(t) <= v <= int_max(t)):
            raise Panic("value %d does not fit %s %s" % (v, t, what))
    return v


def div(a, b):
    if isinstance(a, float) or isinstance(b, float):
        return a / b
    if isinstance(a, RStruct):
        return a._binop("div", b)
    if b == 0:
        raise Panic("division by zero")
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def rem(a, b):
    if isinstance(a, float) or isinstance(b, float):
        import math
        return math.fmod(a, b)
    if b == 0:
        raise Panic("remainder by zero")
    r = abs(a) % abs(b)
    return r if a >= 0 else -r


# ---------------------------------------------------------------- references
class Cell:
    """A boxed scalar local whose address is taken (`&mut x`)."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v

    def get(self):
        return self.v

    def set(self, v):
        self.v = v


class RRef:
    """&mut to an element / field holding a scalar."""
    __slots__ = ("c", "k", "attr")

    def __init__(self, c, k, attr=False):
        self.c, self.k, self.attr = c, k, attr

    def get(self):
        return getattr(self.c, self.k) if self.attr else self.c[self.k]

    def set(self, v):
        if self.attr:
            setattr(self.c, self.k, v)
        else:
            self.c[self.k] = v

    # a reference to a struct element: field access goes through to the struct
    def __getattr__(self, name):
        return getattr(self.get(), name)

    def __setattr__(self, name, v):
        if name in ("c", "k", "attr"):
            object.__setattr__(self, name, v)
        else:
            setattr(self.get(), name, v)


_REFS = (Cell, RRef)


def deref(v):
    if type(v) in _REFS:
        return v.get()
    if type(v) is RPtr:
        return v.b[v.i] if v.i >= 0 else _oob(v)
    return v


def _oob(p):
    raise Panic("raw pointer read before its allocation (%d)" % p.i)


def store(r, v):
    t = type(r)
    if t in _REFS:
        r.set(v)
    elif t is RPtr:
        if r.i < 0:
            _oob(r)
        r.b[r.i] = v
    elif t is MaybeUninitSlot:
        r.write(v)
    elif isinstance(r, RStruct) and type(v) is t:
        # `*r = v` through a reference to a struct: references to structs ARE the struct objects
        # (refmut_local), so the assignment replaces the fields in place
        for f in r._fields:
            setattr(r, f, getattr(v, f))
    else:
        raise Panic("store through non-reference %r" % (r,))


def is_scalar(v):
    return isinstance(v, (int, float)) or v is None


def refmut_local(cell):
    """`&mut x` for a boxed local."""
    v = cell.v
    return cell if is_scalar(v) or isinstance(v, tuple) else v


def refmut_index(c, i):
    v = c[i]
    return RRef(c, i) if is_scalar(v) else v


def refmut_field(o, name):
    v = getattr(o, name)
    return RRef(o, name, True) if is_scalar(v) else v


# ---------------------------------------------------------------- ranges
class RRange:
    __slots__ = ("lo", "hi")

    def __init__(self, lo, hi, incl=False):
        self.lo = lo
        self.hi = (hi + 1) if (incl and hi is not None) else hi

    def __iter__(self):
        if self.hi is None:
            import itertools
            return itertools.count(self.lo)
        return iter(range(self.lo or 0, self.hi))

    def contains(self, x):
        x = deref(x)
        return (self.lo is None or self.lo <= x) and (self.hi is None or x < self.hi)

    def rev(self):
        return RIter(reversed(range(self.lo or 0, self.hi)))

    def step_by(self, n):
        return RIter(iter(range(self.lo or 0, self.hi, n)))

    def len(self):
        return max(0, self.hi - (self.lo or 0))

    def __getattr__(self, name):  # every other iterator adaptor
        return getattr(RIter(iter(self)), name)


# ---------------------------------------------------------------- slices
class RSlice:
    __slots__ = ("b", "o", "n", "arr")

    def __init__(self, b, o=0, n=None, arr=False):
        self.b = b
        self.o = o
        self.n = len(b) - o if n is None else n
        self.arr = arr

    # -- element access
    def __getitem__(self, i):
        if type(i) is int:
            if 0 <= i < self.n:
                return self.b[self.o + i]
            raise Panic("index out of bounds: the len is %d but the index is %d" % (self.n, i))
        if type(i) is RRange:
            lo = i.lo or 0
            hi = self.n if i.hi is None else i.hi
            if not (0 <= lo <= hi <= self.n):
                raise Panic("range %d..%d out of range for slice of length %d" % (lo, hi, self.n))
            return RSlice(self.b, self.o + lo, hi - lo)
        if type(i) is bool:
            return self[int(i)]
        raise Panic("bad index %r" % (i,))

    def __setitem__(self, i, v):
        if type(i) is int and 0 <= i < self.n:
            self.b[self.o + i] = v
        elif type(i) is RRange:
            dst = self[i]
            src = v
            if dst.n != src.n:
                raise Panic("slice assign length")
            dst.b[dst.o:dst.o + dst.n] = src.b[src.o:src.o + src.n]
        else:
            raise Panic("index out of bounds: the len is %d but the index is %r" % (self.n, i))

    def __len__(self):
        return self.n

    def __iter__(self):
        return iter(self.b[self.o:self.o + self.n])

    def __eq__(self, other):
        if isinstance(other, RSlice):
            return self.tolist() == other.tolist()
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def tolist(self):
        return [x.tolist() if isinstance(x, RSlice) else x for x in self.b[self.o:self.o + self.n]]

    def __repr__(self):
        return "RSlice(%r)" % (self.tolist(),)

    def _copy(self):
        return RSlice([x._copy() if isinstance(x, (RSlice, RStruct)) and getattr(x, "arr", True) else x
                       for x in self.b[self.o:self.o + self.n]], 0, self.n, self.arr)

    # -- slice API
    def len(self):
        return self.n

    def is_empty(self):
        return self.n == 0

    def iter(self):
        return RIter(iter(self.b[self.o:self.o + self.n]))

    into_iter = iter

    def iter_mut(self):
        def gen():
            for i in range(self.n):
                v = self.b[self.o + i]
                yield RRef(self, i) if is_scalar(v) or isinstance(v, (tuple, RStruct, REnum)) else v
        return RIter(gen())

    def as_ptr(self):
        return RPtr(self.b, self.o)

    as_mut_ptr = as_ptr

    def as_slice(self):
        return self

    as_mut_slice = as_slice
    as_ref = as_slice
    as_mut = as_slice
    borrow = as_slice
    to_owned = _copy
    clone = _copy

    def to_vec(self):
        return RSlice(list(self.b[self.o:self.o + self.n]))

    def first(self):
        return Some(self[0]) if self.n else NONE

    def last(self):
        return Some(self[self.n - 1]) if self.n else NONE

    def get(self, i):
        if type(i) is RRange:
            try:
                return Some(self[i])
            except Panic:
                return NONE
        return Some(self[i]) if 0 <= i < self.n else NONE

    def get_mut(self, i):
        return Some(refmut_index(self, i)) if 0 <= i < self.n else NONE

    def get_unchecked(self, i):
        return self[i]

    def get_unchecked_mut(self, i):
        return refmut_index(self, i)

    def copy_from_slice(self, src):
        if src.n != self.n:
            raise Panic("copy_from_slice: source slice length (%d) does not match destination slice length (%d)"
                        % (src.n, self.n))
        self.b[self.o:self.o + self.n] = src.b[src.o:src.o + src.n]

    clone_from_slice = copy_from_slice

    def fill(self, v):
        for i in range(self.n):
            self.b[self.o + i] = v

    def chunks(self, k):
        return RIter(RSlice(self.b, self.o + s, min(k, self.n - s)) for s in range(0, self.n, k))

    chunks_mut = chunks

    def chunks_exact(self, k):
        return RIter(RSlice(self.b, self.o + s, k) for s in range(0, self.n - k + 1, k))

    chunks_exact_mut = chunks_exact

    def windows(self, k):
        return RIter(RSlice(self.b, self.o + s, k) for s in range(0, self.n - k + 1))

    def split_at(self, k):
        if k > self.n:
            raise Panic("split_at: mid > len")
        return (RSlice(self.b, self.o, k), RSlice(self.b, self.o + k, self.n - k))

    split_at_mut = split_at

    def reverse(self):
        self.b[self.o:self.o + self.n] = self.b[self.o:self.o + self.n][::-1]

    def swap(self, i, j):
        a, b = self[i], self[j]
        self[i], self[j] = b, a

    def contains(self, x):
        x = deref(x)
        return x in self.b[self.o:self.o + self.n]

    def sort(self):
        self.b[self.o:self.o + self.n] = sorted(self.b[self.o:self.o + self.n])

    sort_unstable = sort

    def sort_by_key(self, f):
        self.b[self.o:self.o + self.n] = sorted(self.b[self.o:self.o + self.n], key=f)  # stable, like Rust

    sort_unstable_by_key = sort_by_key

    def binary_search(self, x):
        x = deref(x)
        lo, hi = 0, self.n
        while lo < hi:
            mid = (lo + hi) // 2
            v = self[mid]
            if v == x:
                return Ok(mid)
            if v < x:
                lo = mid + 1
            else:
                hi = mid
        return Err(lo)

    def concat(self):
        out = []
        for x in self:
            out.extend(x.tolist())
        return RSlice(out)

    # -- Vec API (only valid when the view covers the whole backing list)
    def _whole(self):
        if self.o != 0 or self.n != len(self.b):
            raise Panic("Vec operation on a sub-slice")

    def push(self, v):
        self._whole()
        self.b.append(v)
        self.n += 1

    def pop(self):
        self._whole()
        if not self.n:
            return NONE
        self.n -= 1
        return Some(self.b.pop())

    def extend(self, it):
        self._whole()
        self.b.extend(list(into_iter(it)))
        self.n = len(self.b)

    extend_from_slice = extend

    def truncate(self, k):
        self._whole()
        del self.b[k:]
        self.n = len(self.b)

    def clear(self):
        self.truncate(0)

    def resize(self, k, v):
        self._whole()
        if k < self.n:
            del self.b[k:]
        else:
            self.b.extend([v] * (k - self.n))
        self.n = len(self.b)

    def capacity(self):
        return self.n

    def reserve(self, k):
        pass

    def into_boxed_slice(self):
        return self

    def write(self, v):  # MaybeUninit<[T;N]>::write
        self.copy_from_slice(v)

    def assume_init(self):
        return self

    assume_init_ref = assume_init
    assume_init_mut = assume_init


def repeat(v, n):
    if isinstance(v, (RSlice, RStruct)):
        return RSlice([v._copy() for _ in range(n)], 0, n, True)
    return RSlice([v] * n, 0, n, True)


def array(*els):
    return RSlice(list(els), 0, len(els), True)


def copy_if_array(v):
    """By-value use of an array / Copy-struct place."""
    if type(v) is RSlice:
        return v._copy() if v.arr else v
    if isinstance(v, RStruct) and v._is_copy:
        return v._copy()
    return v


class Aligned:
    """util::Aligned<T> { data: T }"""
    __slots__ = ("data",)

    def __init__(self, data):
        self.data = data


def augop(cur, name, rv, fallback):
    """`cur op= rv` where cur may be a struct implementing the *Assign trait"""
    if isinstance(cur, RStruct):
        if cur._crate.find_method(cur._rname, name + "_assign") is not None:
            cur._crate.call_method(cur, name + "_assign", (rv,))
            return cur
        return cur._binop(name, rv)
    return fallback(cur, rv)


def mwrite(c, i, v):
    c[i] = v
    return RRef(c, i)


class MaybeUninitSlot:
    """Element reference of a `[MaybeUninit<T>]` (r.write(v))."""
    __slots__ = ("c", "k")

    def __init__(self, c, k):
        self.c, self.k = c, k

    def write(self, v):
        self.c[self.k] = v


# ---------------------------------------------------------------- raw pointers
class RPtr:
    __slots__ = ("b", "i")

    def __init__(self, b, i):
        self.b, self.i = b, i

    def add(self, n):
        return RPtr(self.b, self.i + n)

    offset = add
    wrapping_add = add
    wrapping_offset = add

    def sub(self, n):
        return RPtr(self.b, self.i - n)

    def cast(self):
        return self

    def is_null(self):
        return self.b is None

    def read(self):
        return deref(self)

    def write(self, v):
        store(self, v)

    read_unaligned = read
    write_unaligned = write

    def __eq__(self, o):
        return isinstance(o, RPtr) and self.b is o.b and self.i == o.i

    __hash__ = None


def slice_from_raw_parts(p, n):
    return RSlice(p.b, p.i, n)


# ---------------------------------------------------------------- Option / Result / enums
class REnum:
    __slots__ = ("ty", "var", "disc", "p")

    def __init__(self, ty, var, disc=0, p=()):
        self.ty, self.var, self.disc, self.p = ty, var, disc, p

    def __eq__(self, o):
        return isinstance(o, REnum) and self.ty == o.ty and self.var == o.var and self.p == o.p

    def __ne__(self, o):
        return not self.__eq__(o)

    def __hash__(self):
        return hash((self.ty, self.var))

    def __lt__(self, o):
        return (self.disc, self.p) < (o.disc, o.p)

    def __le__(self, o):
        return (self.disc, self.p) <= (o.disc, o.p)

    def __gt__(self, o):
        return (self.disc, self.p) > (o.disc, o.p)

    def __ge__(self, o):
        return (self.disc, self.p) >= (o.disc, o.p)

    def __repr__(self):
        return "%s::%s%s" % (self.ty, self.var, self.p if self.p else "")

    def __index__(self):
        return self.disc

    def _copy(self):
        return self

    def clone(self):
        return self

    # Option / Result API
    def _some(self):
        return self.var in ("Some", "Ok")

    def unwrap(self):
        if not self._some():
            raise Panic("called unwrap() on %s" % self.var)
        return self.p[0]

    def expect(self, msg):
        if not self._some():
            raise Panic(msg)
        return self.p[0]

    unwrap_unchecked = unwrap

    def unwrap_or(self, d):
        return self.p[0] if self._some() else d

    def unwrap_or_else(self, f):
        return self.p[0] if self._some() else f()

    def unwrap_or_default(self):
        return self.p[0] if self._some() else 0

    def is_some(self):
        return self.var == "Some"

    def is_none(self):
        return self.var == "None"

    def is_ok(self):
        return self.var == "Ok"

    def is_err(self):
        return self.var == "Err"

    def ok(self):
        return Some(self.p[0]) if self.var == "Ok" else NONE

    def map(self, f):
        if self.var == "Some":
            return Some(f(self.p[0]))
        if self.var == "Ok":
            return Ok(f(self.p[0]))
        return self

    def map_or(self, d, f):
        return f(self.p[0]) if self._some() else d

    def and_then(self, f):
        return f(self.p[0]) if self._some() else self

    def filter(self, f):
        return self if self._some() and f(self.p[0]) else NONE

    def or_(self, o):
        return self if self._some() else o

    def as_ref(self):
        return self

    as_mut = as_ref
    as_deref = as_ref
    copied = as_ref
    cloned = as_ref

    def iter(self):
        return RIter(iter(self.p[:1] if self._some() else ()))

    into_iter = iter

    def take(self):
        raise Panic("Option::take needs a place")

    def get(self):  # NonZero wrappers never reach here; kept for symmetry
        return self.p[0]


NONE = REnum("Option", "None", 0)


def Some(v):
    return REnum("Option", "Some", 1, (v,))


def Ok(v):
    return REnum("Result", "Ok", 0, (v,))


def Err(v):
    return REnum("Result", "Err", 1, (v,))


# ---------------------------------------------------------------- structs
class RStruct:
    _fields = ()
    _is_copy = False
    _rname = "?"
    _crate = None

    def __init__(self, *a, **kw):
        for f, v in zip(self._fields, a):
            setattr(self, f, v)
        for f, v in kw.items():
            setattr(self, f, v)

    def _copy(self):
        o = object.__new__(type(self))
        for f in self._fields:
            v = getattr(self, f)
            if isinstance(v, RSlice) and v.arr:
                v = v._copy()
            elif isinstance(v, RStruct) and v._is_copy:
                v = v._copy()
            setattr(o, f, v)
        return o

    clone = _copy

    def __eq__(self, o):
        return type(o) is type(self) and all(getattr(self, f) == getattr(o, f) for f in self._fields)

    def __ne__(self, o):
        return not self.__eq__(o)

    def __hash__(self):
        return hash(tuple(getattr(self, f) if not isinstance(getattr(self, f), RSlice) else None
                          for f in self._fields))

    def _key(self):
        return tuple(getattr(self, f) for f in self._fields)

    def __lt__(self, o):
        return self._key() < o._key()

    def __le__(self, o):
        return self._key() <= o._key()

    def __gt__(self, o):
        return self._key() > o._key()

    def __ge__(self, o):
        return self._key() >= o._key()

    def __repr__(self):
        return "%s{%s}" % (self._rname, ", ".join("%s: %r" % (f, getattr(self, f, "?")) for f in self._fields))

    def _binop(self, name, o):
        return self._crate.call_method(self, name, (o,))

    def __mul__(self, o):
        return self._binop("mul", o)

    def __add__(self, o):
        return self._binop("add", o)

    def __sub__(self, o):
        return self._binop("sub", o)

    def __neg__(self):
        return self._crate.call_method(self, "neg", ())

    def __lshift__(self, o):
        return self._binop("shl", o)

    def __rshift__(self, o):
        return self._binop("shr", o)

    def __getitem__(self, i):
        return self._crate.call_method(self, "index", (i,))

    def __setitem__(self, i, v):
        r = self._crate.call_method(self, "index_mut", (i,))
        store(r, v)


# ---------------------------------------------------------------- iterators
def into_iter(x, mut=False):
    t = type(x)
    if t is RIter:
        return x.it
    if t is RSlice:
        return x.iter_mut().it if mut else iter(x)
    if t is RRange:
        return iter(x)
    if t is REnum:
        return x.iter().it
    if hasattr(x, "into_iter"):
        return into_iter(x.into_iter())
    if hasattr(x, "__iter__"):
        return iter(x)
    raise Panic("not iterable: %r" % (x,))


class RIter:
    __slots__ = ("it", "_peek")

    def __init__(self, it):
        self.it = iter(it)

    def __iter__(self):
        return self.it

    def iter(self):
        return self

    into_iter = iter
    by_ref = iter

    def next(self):
        for v in self.it:
            return Some(v)
        return NONE

    def nth(self, n):
        for _ in range(n):
            next(self.it, None)
        return self.next()

    def map(self, f):
        return RIter(map(f, self.it))

    def for_each(self, f):
        for v in self.it:
            f(v)

    def zip(self, o, mut=False):
        return RIter(zip(self.it, into_iter(o, mut)))

    def enumerate(self):
        return RIter(enumerate(self.it))

    def take(self, n):
        import itertools
        return RIter(itertools.islice(self.it, n))

    def skip(self, n):
        import itertools
        return RIter(itertools.islice(self.it, n, None))

    def step_by(self, n):
        import itertools
        return RIter(itertools.islice(self.it, 0, None, n))

    def rev(self):
        return RIter(reversed(list(self.it)))

    def chain(self, o):
        import itertools
        return RIter(itertools.chain(self.it, into_iter(o)))

    def filter(self, f):
        return RIter(v for v in self.it if f(v))

    def filter_map(self, f):
        def gen():
            for v in self.it:
                r = f(v)
                if r.var == "Some":
                    yield r.p[0]
        return RIter(gen())

    def flat_map(self, f):
        def gen():
            for v in self.it:
                for w in into_iter(f(v)):
                    yield w
        return RIter(gen())

    def flatten(self):
        return self.flat_map(lambda v: v)

    def take_while(self, f):
        import itertools
        return RIter(itertools.takewhile(f, self.it))

    def skip_while(self, f):
        import itertools
        return RIter(itertools.dropwhile(f, self.it))

    def copied(self):
        return RIter(deref(v) for v in self.it)

    cloned = copied

    def peekable(self):
        return self

    def fuse(self):
        return self

    def cycle(self):
        import itertools
        return RIter(itertools.cycle(list(self.it)))

    def sum(self):
        s = 0
        for v in self.it:
            s = s + deref(v)
        return s

    def product(self):
        s = 1
        for v in self.it:
            s = s * deref(v)
        return s

    def count(self):
        return sum(1 for _ in self.it)

    def len(self):
        l = list(self.it)
        self.it = iter(l)
        return len(l)

    def last(self):
        r = NONE
        for v in self.it:
            r = Some(v)
        return r

    def fold(self, init, f):
        acc = init
        for v in self.it:
            acc = f(acc, v)
        return acc

    def all(self, f):
        return all(f(v) for v in self.it)

    def any(self, f):
        return any(f(v) for v in self.it)

    def position(self, f):
        for i, v in enumerate(self.it):
            if f(v):
                return Some(i)
        return NONE

    def rposition(self, f):
        l = list(self.it)
        for i in range(len(l) - 1, -1, -1):
            if f(l[i]):
                return Some(i)
        return NONE

    def find(self, f):
        for v in self.it:
            if f(v):
                return Some(v)
        return NONE

    def find_map(self, f):
        for v in self.it:
            r = f(v)
            if r.var == "Some":
                return r
        return NONE

    # Rust: max / max_by_key return the LAST maximum, min / min_by_key the FIRST minimum
    def max(self):
        best = None
        first = True
        for v in self.it:
            v = deref(v)
            if first or v >= best:
                best, first = v, False
        return NONE if first else Some(best)

    def min(self):
        best = None
        first = True
        for v in self.it:
            v = deref(v)
            if first or v < best:
                best, first = v, False
        return NONE if first else Some(best)

    def max_by_key(self, f):
        best = bk = None
        first = True
        for v in self.it:
            k = f(v)
            if first or k >= bk:
                best, bk, first = v, k, False
        return NONE if first else Some(best)

    def min_by_key(self, f):
        best = bk = None
        first = True
        for v in self.it:
            k = f(v)
            if first or k < bk:
                best, bk, first = v, k, False
        return NONE if first else Some(best)

    def collect(self):
        return RSlice(list(self.it))

    def unzip(self):
        a, b = [], []
        for x, y in self.it:
            a.append(x)
            b.append(y)
        return (RSlice(a), RSlice(b))


def izip(*its):
    return RIter(zip(*its))


# ---------------------------------------------------------------- integer methods
def _need(t, name):
    if t is None:
        raise Panic("rustlite: method %s needs the receiver's integer type (add an annotation hint)" % name)
    return t


def msb(x):
    if x <= 0:
        raise Panic("msb of %d" % x)
    return x.bit_length() - 1


def round_shift(v, bit):
    return (v + ((1 << bit) >> 1)) >> bit


def clamp3(v, lo, hi):
    if lo > hi:
        raise Panic("clamp: min > max")
    return lo if v < lo else hi if v > hi else v


def int_method(v, t, name, a):
    """v.name(*a) for an integer v of static type t (t may be None when irrelevant)."""
    if name == "min":
        b = deref(a[0])
        return v if v <= b else b
    if name == "max":
        b = deref(a[0])
        return b if b >= v else v
    if name == "clamp":
        return clamp3(v, a[0], a[1])
    if name == "abs":
        return -v if v < 0 else v
    if name == "unsigned_abs":
        return -v if v < 0 else v
    if name == "abs_diff":
        return abs(v - a[0])
    if name == "signum":
        return (v > 0) - (v < 0)
    if name == "pow":
        return v ** a[0]
    if name in ("get", "into", "clone", "to_owned", "as_", "borrow", "to_asm_stride"):
        return v
    if name in ("to_usize", "to_i32", "to_u32", "to_u16", "to_i16", "to_u8", "to_u64", "to_i64", "to_isize"):
        t2 = name[3:]          # num_traits::ToPrimitive
        return Some(v) if int_min(t2) <= v <= int_max(t2) else NONE
    if name == "try_into":
        # the target type is inferred in Rust; every use in the reference goes to an unsigned type
        return Ok(v) if v >= 0 else Err(None)
    if name == "is_power_of_two":
        return v > 0 and (v & (v - 1)) == 0
    if name == "next_power_of_two":
        return 1 if v <= 1 else 1 << (v - 1).bit_length()
    if name == "ilog" and not a:
        # v_frame::math::ILog: floor(log2(x)) + 1, and 0 for x <= 0
        return v.bit_length() if v > 0 else 0
    if name == "ilog2" or name == "ilog":
        if v <= 0:
            raise Panic("ilog2 of %d" % v)
        if name == "ilog":
            r, b = 0, a[0]
            while v >= b:
                v //= b
                r += 1
            return r
        return v.bit_length() - 1
    if name == "trailing_zeros":
        if v == 0:
            return INT_BITS[_need(t, name)]
        return (v & -v).bit_length() - 1
    if name == "leading_zeros":
        b = INT_BITS[_need(t, name)]
        return b - (v & ((1 << b) - 1)).bit_length()
    if name == "count_ones":
        b = INT_BITS[_need(t, name)]
        return bin(v & ((1 << b) - 1)).count("1")
    if name == "div_ceil":
        return -((-v) // a[0])
    if name == "div_euclid":
        return v // a[0] if a[0] > 0 else -(v // -a[0])
    if name == "rem_euclid":
        return v % abs(a[0])
    if name.startswith("wrapping_"):
        op = name[9:]
        t = _need(t, name)
        r = {"add": lambda: v + a[0], "sub": lambda: v - a[0], "mul": lambda: v * a[0],
             "neg": lambda: -v, "shl": lambda: v << (a[0] % INT_BITS[t]),
             "shr": lambda: v >> (a[0] % INT_BITS[t]), "abs": lambda: abs(v)}[op]()
        return wrap(r, t)
    if name.startswith("saturating_"):
        op = name[11:]
        t = _need(t, name)
        r = {"add": lambda: v + a[0], "sub": lambda: v - a[0], "mul": lambda: v * a[0],
             "pow": lambda: v ** a[0]}[op]()
        return max(int_min(t), min(int_max(t), r))
    if name.startswith("checked_"):
        op = name[8:]
        t = _need(t, name)
        if op in ("div", "rem") and a[0] == 0:
            return NONE
        r = {"add": lambda: v + a[0], "sub": lambda: v - a[0], "mul": lambda: v * a[0],
             "div": lambda: div(v, a[0]), "rem": lambda: rem(v, a[0]),
             "shl": lambda: v << a[0], "shr": lambda: v >> a[0]}[op]()
        return Some(r) if int_min(t) <= r <= int_max(t) else NONE
    if name.startswith("overflowing_"):
        op = name[12:]
        t = _need(t, name)
        r = {"add": lambda: v + a[0], "sub": lambda: v - a[0], "mul": lambda: v * a[0]}[op]()
        return (wrap(r, t), not (int_min(t) <= r <= int_max(t)))
    if name == "align_power_of_two":  # v_frame::math::Fixed
        n = a[0]
        return (v + (1 << n) - 1) & ~((1 << n) - 1)
    if name == "align_power_of_two_and_shift":
        n = a[0]
        return (v + (1 << n) - 1) >> n
    if name == "floor_log2":
        n = a[0]
        return v & ~((1 << n) - 1)
    if name == "cmp":
        b = deref(a[0])
        return REnum("Ordering", "Less" if v < b else "Greater" if v > b else "Equal", (v > b) - (v < b))
    if name == "partial_cmp":
        b = deref(a[0])
        return Some(REnum("Ordering", "Less" if v < b else "Greater" if v > b else "Equal", (v > b) - (v < b)))
    if name == "then" and isinstance(v, bool):
        return Some(a[0]()) if v else NONE
    if name == "then_some" and isinstance(v, bool):
        return Some(a[0]) if v else NONE
    if name == "not":
        return not v
    if name == "to_bits":
        return v
    if name == "swap_bytes":
        b = INT_BITS[_need(t, name)] // 8
        return int.from_bytes((v & ((1 << 8 * b) - 1)).to_bytes(b, "little"), "big")
    if name == "mul_add":
        # f64::mul_add is FUSED (one rounding): exact rational product + sum, rounded once
        from fractions import Fraction
        import math
        if all(isinstance(x, (int, float)) and math.isfinite(x) for x in (v, a[0], a[1])):
            return float(Fraction(v) * Fraction(a[0]) + Fraction(a[1]))
        return v * a[0] + a[1]
    if name in ("sqrt", "floor", "ceil", "round", "ln", "log2", "exp", "exp2", "powf", "powi", "log10",
                "is_nan", "is_finite", "trunc", "recip", "cbrt"):
        import math
        x = float(v)
        if name == "round":
            return math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1)
        if name == "powf" or name == "powi":
            return x ** a[0]
        if name == "is_nan":
            return x != x
        if name == "is_finite":
            return math.isfinite(x)
        if name == "recip":
            return 1.0 / x
        if name == "ln":
            return math.log(x)
        if name == "cbrt":
            return math.copysign(abs(x) ** (1.0 / 3.0), x)
        return float(getattr(math, name)(x))
    raise Panic("rustlite: integer method %s not implemented" % name)


def tuple_method(v, name, a):
    if name in ("clone", "into", "to_owned"):
        return v
    if name == "cmp":
        b = a[0]
        return REnum("Ordering", "Less" if v < b else "Greater" if v > b else "Equal", (v > b) - (v < b))
    raise Panic("rustlite: tuple method %s" % name)


def try_from(v, t):
    v = deref(v)
    return Ok(v) if int_min(t) <= v <= int_max(t) else Err(None)


# ---------------------------------------------------------------- v_frame stand-ins
# v_frame 0.3.9 is a third-party crate that is NOT under /root/reference (SURVEY.md
# section 8c).  These few types restate its documented layout: element (x, y) of a
# plane lives at data[(yorigin + y) * stride + xorigin + x].
class PlaneConfig(RStruct):
    _fields = ("stride", "alloc_height", "width", "height", "xdec", "ydec", "xpad", "ypad", "xorigin", "yorigin")
    _rname = "PlaneConfig"
    _is_copy = False


def _plane_config_new(width, height, xdec, ydec, xpad, ypad, type_size):
    # v_frame PlaneConfig::new: xorigin and stride aligned to 64 BYTES (see Plane.new)
    al = 64 // type_size
    xorigin = _align(xpad, al)
    stride = _align(xorigin + width + xpad, al)
    return PlaneConfig(stride, ypad + height + ypad, width, height, xdec, ydec, xpad, ypad, xorigin, ypad)


PlaneConfig.new = staticmethod(_plane_config_new)


class PlaneOffset(RStruct):
    _fields = ("x", "y")
    _rname = "PlaneOffset"
    _is_copy = True


def _align(v, a):
    return (v + a - 1) // a * a


class PlaneData(list):
    """Plane::data (a PlaneData<T> that derefs to [T]) where reference code slices it directly
    (`&plane.data[..]`): a list that also takes Rust ranges"""

    def __getitem__(self, i):
        if type(i) is RRange:
            lo = i.lo or 0
            hi = len(self) if i.hi is None else i.hi
            return RSlice(self, lo, hi - lo)
        return list.__getitem__(self, i)


class Plane(RStruct):
    _fields = ("data", "cfg")
    _rname = "Plane"

    @staticmethod
    def new(width, height, xdec, ydec, xpad, ypad, bpp=1):
        # v_frame PlaneConfig::new: xorigin and stride aligned to 64 BYTES
        al = 64 // bpp
        xorigin = _align(xpad, al)
        yorigin = ypad
        stride = _align(xorigin + width + xpad, al)
        alloc_height = yorigin + height + ypad
        cfg = PlaneConfig(stride, alloc_height, width, height, xdec, ydec, xpad, ypad, xorigin, yorigin)
        fill = 128 if bpp == 1 else 0  # contents are always overwritten by the generators
        return Plane([fill] * (stride * alloc_height), cfg)

    @staticmethod
    def from_slice(data, stride):
        # v_frame 0.3.9 Plane::from_slice: a plane of width = stride, height = len / stride, no
        # padding, holding a copy of the data (used by the reference's unit tests)
        vals = list(data)
        if stride <= 0 or len(vals) % stride:
            raise Panic("Plane::from_slice: %d elements, stride %d" % (len(vals), stride))
        h = len(vals) // stride
        return Plane(PlaneData(vals), PlaneConfig(stride, h, stride, h, 0, 0, 0, 0, 0, 0))

    def slice(self, po):
        return PlaneSlice(self, po.x, po.y)

    def mut_slice(self, po):
        return PlaneSlice(self, po.x, po.y)

    def as_region(self):
        return self._region(0, 0, self.cfg.width, self.cfg.height)

    as_region_mut = as_region

    def region(self, area):
        # src/frame/plane.rs:24-32: the parent rectangle is the allocation right / below the origin;
        # PlaneRegion::from_slice's asserts (plane_region.rs:166-169)
        c = self.cfg
        r = area_to_rect(area, c.xdec, c.ydec, c.stride - c.xorigin, c.alloc_height - c.yorigin)
        if not (r[0] >= -c.xorigin and r[1] >= -c.yorigin and c.xorigin + r[0] + r[2] <= c.stride and
                c.yorigin + r[1] + r[3] <= c.alloc_height):
            raise Panic("PlaneRegion::from_slice: rectangle %r outside the allocation" % (r,))
        return self._region(r[0], r[1], r[2], r[3])

    region_mut = region

    def _region(self, x, y, w, h):
        return PlaneRegion(self.data, (self.cfg.yorigin + y) * self.cfg.stride + self.cfg.xorigin + x,
                           self.cfg, x, y, w, h)

    def row_range(self, x, y):
        # v_frame Plane::row_range: the indices of row y from column x to the end of the row's allocation
        c = self.cfg
        if c.yorigin + y < 0 or c.xorigin + x < 0 or c.yorigin + y >= c.alloc_height:
            raise Panic("Plane::row_range outside the allocation (x %d y %d)" % (x, y))
        base = (c.yorigin + y) * c.stride + c.xorigin + x
        return RRange(base, base + c.stride - (c.xorigin + x))

    def p(self, x, y):
        return self.data[(self.cfg.yorigin + y) * self.cfg.stride + self.cfg.xorigin + x]

    def setp(self, x, y, v):
        self.data[(self.cfg.yorigin + y) * self.cfg.stride + self.cfg.xorigin + x] = v


class PlaneSlice(RStruct):
    _fields = ("plane", "x", "y")
    _rname = "PlaneSlice"

    def _base(self, row):
        c = self.plane.cfg
        by = c.yorigin + self.y + row
        bx = c.xorigin + self.x
        if by < 0 or by >= c.alloc_height or bx < 0 or bx > c.stride:
            raise Panic("PlaneSlice row out of the allocation (x %d y %d row %d)" % (self.x, self.y, row))
        return by * c.stride + bx, c.stride - bx

    def __getitem__(self, row):
        base, width = self._base(row)
        return RSlice(self.plane.data, base, width)

    def row(self, y):
        # v_frame PlaneSlice::row: row y of the slice, from its x to the end of the row's allocation
        return self[y]

    def as_ptr(self):
        return RPtr(self.plane.data, self._base(0)[0])

    as_mut_ptr = as_ptr

    def go_up(self, n):
        return PlaneSlice(self.plane, self.x, self.y - n)

    def go_down(self, n):
        return PlaneSlice(self.plane, self.x, self.y + n)

    def go_left(self, n):
        return PlaneSlice(self.plane, self.x - n, self.y)

    def go_right(self, n):
        return PlaneSlice(self.plane, self.x + n, self.y)

    def reslice(self, xo, yo):
        return PlaneSlice(self.plane, self.x + xo, self.y + yo)

    def subslice(self, xo, yo):
        return PlaneSlice(self.plane, self.x + xo, self.y + yo)

    def clamp(self):
        # v_frame 0.3.9 PlaneSlice::clamp: x in [-xorigin, width], y in [-yorigin, height]
        c = self.plane.cfg
        return PlaneSlice(self.plane, max(min(self.x, c.width), -c.xorigin),
                          max(min(self.y, c.height), -c.yorigin))

    def rows_iter(self):
        c = self.plane.cfg
        def gen():
            r = 0
            while c.yorigin + self.y + r < c.alloc_height:
                yield self[r]
                r += 1
        return RIter(gen())

    def p(self, add_x, add_y):
        return self[add_y][add_x]

    def accessible(self, add_x, add_y):
        c = self.plane.cfg
        return (self.x + add_x + c.xorigin >= 0 and self.y + add_y + c.yorigin >= 0 and
                self.x + add_x + c.xorigin < c.stride and self.y + add_y + c.yorigin < c.alloc_height)

    def accessible_neg(self, sub_x, sub_y):
        c = self.plane.cfg
        return self.x - sub_x + c.xorigin >= 0 and self.y - sub_y + c.yorigin >= 0


def area_to_rect(area, xdec, ydec, pw, ph):
    """src/tiling/plane_region.rs Area::to_rect (restated for the stand-in)."""
    v = area.var
    f = area.p
    if v == "Rect":
        return (f[0], f[1], f[2], f[3])
    if v == "StartingAt":
        return (f[0], f[1], pw - f[0], ph - f[1])
    if v == "BlockRect":
        bo = f[0]
        return ((bo.x >> xdec) << 2, (bo.y >> ydec) << 2, f[1], f[2])
    if v == "BlockStartingAt":
        bo = f[0]
        x, y = (bo.x >> xdec) << 2, (bo.y >> ydec) << 2
        return (x, y, pw - x, ph - y)
    raise Panic("area %r" % (area,))


class Rect(RStruct):
    _fields = ("x", "y", "width", "height")
    _rname = "Rect"
    _is_copy = True
    # the name `Rect` is also the variant Area::Rect { x, y, width, height }: patterns written
    # `Rect { x, y, .. }` are compiled against that variant's positional payload
    var = "Rect"

    @property
    def p(self):
        return (self.x, self.y, self.width, self.height)


class PlaneRegion(RStruct):
    """src/tiling/plane_region.rs PlaneRegion / PlaneRegionMut (macro-generated there,
    restated here): a rectangle of a plane; region[r] is row r restricted to the rectangle."""
    _fields = ("data", "base", "plane_cfg", "rx", "ry", "rw", "rh")
    _rname = "PlaneRegion"

    def rect(self):
        return Rect(self.rx, self.ry, self.rw, self.rh)

    @staticmethod
    def new(plane, rect):
        return plane._region(rect.x, rect.y, rect.width, rect.height)

    @staticmethod
    def new_from_plane(plane):
        return plane.as_region()

    @staticmethod
    def from_slice(data, cfg, rect):
        # plane_region.rs: a region over a caller-owned buffer laid out by `cfg`
        data = deref(data)
        if isinstance(data, RSlice):
            assert data.o == 0
            data = data.b
        x, y, w, h = rect.p if isinstance(rect, REnum) else (rect.x, rect.y, rect.width, rect.height)
        return PlaneRegion(data, (cfg.yorigin + y) * cfg.stride + cfg.xorigin + x, cfg, x, y, w, h)

    def __getitem__(self, r):
        if not (0 <= r < self.rh):
            raise Panic("PlaneRegion row %d out of %d" % (r, self.rh))
        return RSlice(self.data, self.base + r * self.plane_cfg.stride, self.rw)

    def rows_iter(self):
        s = self.plane_cfg.stride
        return RIter(RSlice(self.data, self.base + r * s, self.rw) for r in range(self.rh))

    rows_iter_mut = rows_iter

    def data_ptr(self):
        return RPtr(self.data, self.base)

    data_ptr_mut = data_ptr

    def subregion(self, area):
        x, y, w, h = area_to_rect(area, self.plane_cfg.xdec, self.plane_cfg.ydec, self.rw, self.rh)
        if not (0 <= x <= self.rw and 0 <= y <= self.rh):
            raise Panic("subregion origin outside the region")
        return PlaneRegion(self.data, self.base + y * self.plane_cfg.stride + x, self.plane_cfg,
                           self.rx + x, self.ry + y, w, h)

    subregion_mut = subregion

    def as_const(self):
        return self

    def vert_windows(self, h):
        n = max(0, self.rh - h + 1)
        s = self.plane_cfg.stride
        return RIter(PlaneRegion(self.data, self.base + k * s, self.plane_cfg, self.rx, self.ry + k, self.rw, h)
                     for k in range(n))

    def horz_windows(self, w):
        n = max(0, self.rw - w + 1)
        return RIter(PlaneRegion(self.data, self.base + k, self.plane_cfg, self.rx + k, self.ry, w, self.rh)
                     for k in range(n))

    def to_frame_block_offset(self, tile_bo):
        # plane_region.rs:349-361
        xdec, ydec = self.plane_cfg.xdec, self.plane_cfg.ydec
        G = self._crate.G
        if "S_PlaneBlockOffset" not in G:
            self._crate.autoload("PlaneBlockOffset")
        bx = self.rx >> (2 - xdec)
        by = self.ry >> (2 - ydec)
        return G["S_PlaneBlockOffset"](G["S_BlockOffset"](x=bx + tile_bo._0.x, y=by + tile_bo._0.y))

    def frame_block_offset(self):
        G = self._crate.G
        if "S_BlockOffset" not in G:
            self._crate.autoload("BlockOffset")
        zero = type("TBO", (), {})()
        zero._0 = G["S_BlockOffset"](x=0, y=0)
        return self.to_frame_block_offset(zero)

    def is_null(self):
        return self.data is None

    def home(self):
        # plane_region.rs:325-338: the same pixels, the rectangle's origin moved to (0, 0)
        return PlaneRegion(self.data, self.base, self.plane_cfg, 0, 0, self.rw, self.rh)

    def scratch_copy(self):
        # plane_region.rs:390-401: a new plane without padding holding the rectangle's pixels
        c = self.plane_cfg
        ret = Plane.new(self.rw, self.rh, c.xdec, c.ydec, 0, 0)
        ret.data = PlaneData(ret.data)
        s = c.stride
        for r in range(self.rh):
            b = (ret.cfg.yorigin + r) * ret.cfg.stride + ret.cfg.xorigin
            ret.data[b:b + self.rw] = list(self.data[self.base + r * s:self.base + r * s + self.rw])
        return ret
