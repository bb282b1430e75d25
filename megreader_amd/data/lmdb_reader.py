"""Read-only LMDB container reader -- what the reference's `data/lmdb_dataset.py:59-88` needs from the
`lmdb` module: `lmdb.open(path, max_dbs=1, lock=False)`, `env.open_db(b'image')`, `env.begin(db=...)`, `txn.get(key)`.

The `lmdb` Python module (a C extension around liblmdb) is not installed in the build image and cannot be, so this
is a pure-Python reader of the documented on-disk format of LMDB 0.9 (`data.mdb`, 64-bit little-endian build -- the
only layout the py-lmdb wheels produce on x86-64 / aarch64 Linux):

  page       = 16-byte header {pgno u64, pad u16, flags u16, lower u16, upper u16 | overflow page count u32}
               followed by the u16 node-offset array (`(lower - 16) / 2` entries) ; flags: BRANCH 1, LEAF 2,
               OVERFLOW 4, META 8, LEAF2 0x20
  meta page  = pages 0 and 1: header + {magic 0xBEEFC0DE, version 1, address u64, mapsize u64, 2 x MDB_db
               (FREE, MAIN), last_pgno u64, txnid u64}; the newer txnid wins; page size = FREE db's `pad` field
  MDB_db     = {pad u32, flags u16, depth u16, branch_pages u64, leaf_pages u64, overflow_pages u64, entries u64,
               root u64} (48 bytes); an empty tree has root = 2^64-1
  node       = {lo u16, hi u16, flags u16, ksize u16, key bytes, data}; leaf: data size = lo | hi << 16 and the data
               follow the key, or (flag BIGDATA 1) an 8-byte overflow page number whose page(s) hold the data
               contiguously after one 16-byte header; branch: child page = lo | hi << 16 | flags << 32, key[0] = -inf
  named DBs  = records of the MAIN db: key = name, data = MDB_db, flag SUBDATA 2
  key order  = memcmp, shorter key first on a tie (the default comparator; the reference sets no custom one)

`write_environment` is the inverse (bulk-loads sorted records into fresh B+tree pages) and exists for tests and for
building fixtures; DUPSORT databases, the free-page list and write transactions are not implemented (the reference
never uses them on this path).

PARITY UNPINNED: no liblmdb build and no LMDB file exists in the build image, so neither function has been checked
against the real library -- the tests check reader and writer against each other and against the constants above.
"""
import builtins
import mmap
import os
import struct

MAGIC = 0xBEEFC0DE
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
PAGEHDRSZ = 16
NODESZ = 8
INVALID = (1 << 64) - 1
_DB = struct.Struct('<IHHQQQQQ')          # MDB_db
_META = struct.Struct('<IIQQ')            # magic, version, address, mapsize


class Error(Exception):
    pass


class _Db(object):
    def __init__(self, raw):
        (self.pad, self.flags, self.depth, self.branch_pages, self.leaf_pages, self.overflow_pages, self.entries,
         self.root) = _DB.unpack(raw)


class Environment(object):
    """`Environment(path)`: `path` is the directory holding `data.mdb` (or the file itself with subdir=False)."""

    def __init__(self, path, subdir=True, **_ignored):
        fname = os.path.join(path, 'data.mdb') if subdir else path
        self._file = builtins.open(fname, 'rb')
        size = os.fstat(self._file.fileno()).st_size
        if size < 2 * 512:
            raise Error('%s: too small to be an LMDB environment' % fname)
        self._map = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
        metas = []
        psize = None
        for i in range(2):
            off = i * (psize or 4096)
            if i == 1 and psize is None:
                raise Error('%s: first meta page unreadable' % fname)
            flags = struct.unpack_from('<H', self._map, off + 10)[0]
            magic, version, _addr, mapsize = _META.unpack_from(self._map, off + PAGEHDRSZ)
            if magic != MAGIC or not (flags & P_META):
                if i == 0:
                    raise Error('%s: not an LMDB environment (magic %#x)' % (fname, magic))
                continue
            if version != 1:
                raise Error('%s: unsupported LMDB data version %d' % (fname, version))
            dbs = [_Db(self._map[off + PAGEHDRSZ + 24 + j * 48: off + PAGEHDRSZ + 24 + (j + 1) * 48]) for j in range(2)]
            last_pg, txnid = struct.unpack_from('<QQ', self._map, off + PAGEHDRSZ + 24 + 96)
            if psize is None:
                psize = dbs[0].pad
                if psize < 512 or psize > 65536 or psize & (psize - 1):
                    raise Error('%s: bad page size %d' % (fname, psize))
            metas.append((txnid, dbs, last_pg, mapsize))
        self.psize = psize
        self.txnid, dbs, self.last_pgno, self.mapsize = max(metas, key=lambda m: m[0])
        self._main = dbs[1]
        self._named = {}

    # -- page access ------------------------------------------------------------------------------------------
    def _page(self, pgno):
        off = pgno * self.psize
        if off + self.psize > len(self._map):
            raise Error('page %d beyond the end of the file' % pgno)
        flags, lower, upper = struct.unpack_from('<HHH', self._map, off + 10)
        return off, flags, lower, upper

    def _node(self, off, idx):
        ptr = struct.unpack_from('<H', self._map, off + PAGEHDRSZ + 2 * idx)[0]
        lo, hi, flags, ksize = struct.unpack_from('<HHHH', self._map, off + ptr)
        return off + ptr, lo, hi, flags, ksize

    def _leaf_value(self, noff, lo, hi, flags, ksize):
        size = lo | (hi << 16)
        doff = noff + NODESZ + ksize
        if flags & F_BIGDATA:
            pgno = struct.unpack_from('<Q', self._map, doff)[0]
            poff, pflags, _, _ = self._page(pgno)
            if not (pflags & P_OVERFLOW):
                raise Error('page %d is not an overflow page' % pgno)
            return bytes(self._map[poff + PAGEHDRSZ: poff + PAGEHDRSZ + size]), flags
        return bytes(self._map[doff: doff + size]), flags

    @staticmethod
    def _cmp(a, b):
        return (a > b) - (a < b)   # bytes compare = memcmp, shorter first on a common prefix

    def _find(self, db, key):
        """(value, node flags) of `key` in tree `db`, or None."""
        if db.root == INVALID:
            return None
        pgno = db.root
        for _ in range(64):
            off, flags, lower, _ = self._page(pgno)
            n = (lower - PAGEHDRSZ) // 2
            if flags & P_BRANCH:
                lo_i, hi_i = 1, n - 1      # key[0] of a branch page is -infinity
                child = 0
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) // 2
                    noff, _, _, _, ks = self._node(off, mid)
                    if self._cmp(bytes(self._map[noff + NODESZ: noff + NODESZ + ks]), key) <= 0:
                        child = mid
                        lo_i = mid + 1
                    else:
                        hi_i = mid - 1
                noff, lo, hi, nflags, _ = self._node(off, child)
                pgno = lo | (hi << 16) | (nflags << 32)
                continue
            if not (flags & P_LEAF) or (flags & P_LEAF2):
                raise Error('unexpected page type %#x at page %d' % (flags, pgno))
            lo_i, hi_i = 0, n - 1
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) // 2
                noff, lo, hi, nflags, ks = self._node(off, mid)
                c = self._cmp(bytes(self._map[noff + NODESZ: noff + NODESZ + ks]), key)
                if c == 0:
                    return self._leaf_value(noff, lo, hi, nflags, ks)
                if c < 0:
                    lo_i = mid + 1
                else:
                    hi_i = mid - 1
            return None
        raise Error('tree deeper than 64 levels: corrupt file')

    def _walk(self, db):
        """(key, value, node flags) of every record of tree `db` in key order."""
        if db.root == INVALID:
            return
        stack = [db.root]
        while stack:
            pgno = stack.pop()
            off, flags, lower, _ = self._page(pgno)
            n = (lower - PAGEHDRSZ) // 2
            if flags & P_BRANCH:
                kids = []
                for i in range(n):
                    _, lo, hi, nflags, _ = self._node(off, i)
                    kids.append(lo | (hi << 16) | (nflags << 32))
                stack.extend(reversed(kids))
            else:
                for i in range(n):
                    noff, lo, hi, nflags, ks = self._node(off, i)
                    val, fl = self._leaf_value(noff, lo, hi, nflags, ks)
                    yield bytes(self._map[noff + NODESZ: noff + NODESZ + ks]), val, fl

    # -- the subset of the lmdb API the reference uses --------------------------------------------------------------
    def open_db(self, key=None, **_ignored):
        if key is None:
            return self._main
        if key not in self._named:
            rec = self._find(self._main, key)
            if rec is None or not (rec[1] & F_SUBDATA) or len(rec[0]) != 48:
                raise Error('named database %r not found' % (key,))
            db = _Db(rec[0])
            if db.flags & 0x04:   # MDB_DUPSORT
                raise Error('DUPSORT databases are not supported by this reader')
            self._named[key] = db
        return self._named[key]

    def begin(self, db=None, write=False, **_ignored):
        if write:
            raise Error('read-only reader')
        return Transaction(self, db if db is not None else self._main)

    def stat(self):
        m = self._main
        return {'psize': self.psize, 'depth': m.depth, 'branch_pages': m.branch_pages, 'leaf_pages': m.leaf_pages,
                'overflow_pages': m.overflow_pages, 'entries': m.entries}

    def close(self):
        if self._map is not None:
            self._map.close()
            self._file.close()
            self._map = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Transaction(object):
    def __init__(self, env, db):
        self._env, self._db = env, db

    def get(self, key, default=None):
        rec = self._env._find(self._db, bytes(key))
        return default if rec is None else rec[0]

    def cursor(self):
        return ((k, v) for k, v, _ in self._env._walk(self._db))

    def stat(self):
        d = self._db
        return {'depth': d.depth, 'branch_pages': d.branch_pages, 'leaf_pages': d.leaf_pages,
                'overflow_pages': d.overflow_pages, 'entries': d.entries}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def open(path, **kwargs):  # noqa: A001 - mirrors lmdb.open
    """`lmdb.open(path, max_dbs=..., lock=False, readonly=True)` look-alike (read-only; the extra keywords are accepted
    and ignored)."""
    return Environment(path, subdir=kwargs.pop('subdir', True), **kwargs)


# ----------------------------------------------------------------------------------------------------------------
# writer (tests / fixtures)
# ----------------------------------------------------------------------------------------------------------------
class _PageWriter(object):
    def __init__(self, psize):
        self.psize = psize
        self.pages = {}          # pgno -> bytes (possibly several pages long for overflow runs)
        self.next = 2
        self.stats = {'branch': 0, 'leaf': 0, 'overflow': 0}

    def alloc(self, count=1):
        pg = self.next
        self.next += count
        return pg

    def _emit(self, pgno, flags, nodes):
        """nodes: list of node byte strings; lays them out from the end of the page like mdb_node_add."""
        buf = bytearray(self.psize)
        upper = self.psize
        ptrs = []
        for nd in nodes:
            sz = (len(nd) + 1) & ~1   # nodes are 2-byte aligned
            upper -= sz
            buf[upper: upper + len(nd)] = nd
            ptrs.append(upper)
        lower = PAGEHDRSZ + 2 * len(nodes)
        if lower > upper:
            raise Error('page overflow while writing')
        struct.pack_into('<QHHHH', buf, 0, pgno, 0, flags, lower, upper)
        for i, p in enumerate(ptrs):
            struct.pack_into('<H', buf, PAGEHDRSZ + 2 * i, p)
        self.pages[pgno] = bytes(buf)

    def build_tree(self, records, sub_flags=None):
        """records: sorted [(key, value)] -> _Db fields (depth, branch, leaf, overflow, entries, root)."""
        if not records:
            return (0, 0, 0, 0, 0, INVALID)
        nodemax = (((self.psize - PAGEHDRSZ) // 2) & ~1) - 2
        b0, l0, o0 = self.stats['branch'], self.stats['leaf'], self.stats['overflow']
        level = []     # (first key, pgno) of the pages of the current level
        cur, cur_bytes, first = [], 0, None
        for idx, (key, val) in enumerate(records):
            flags = 0 if sub_flags is None else sub_flags[idx]
            if NODESZ + len(key) + len(val) > nodemax:
                npg = (PAGEHDRSZ + len(val) + self.psize - 1) // self.psize
                opg = self.alloc(npg)
                buf = bytearray(npg * self.psize)
                struct.pack_into('<QHHI', buf, 0, opg, 0, P_OVERFLOW, npg)
                buf[PAGEHDRSZ: PAGEHDRSZ + len(val)] = val
                self.pages[opg] = bytes(buf)
                self.stats['overflow'] += npg
                node = struct.pack('<HHHH', len(val) & 0xFFFF, len(val) >> 16, flags | F_BIGDATA, len(key)) + key + \
                    struct.pack('<Q', opg)
            else:
                node = struct.pack('<HHHH', len(val) & 0xFFFF, len(val) >> 16, flags, len(key)) + key + val
            need = ((len(node) + 1) & ~1) + 2
            if cur and cur_bytes + need > self.psize - PAGEHDRSZ:
                pg = self.alloc()
                self._emit(pg, P_LEAF, cur)
                self.stats['leaf'] += 1
                level.append((first, pg))
                cur, cur_bytes, first = [], 0, None
            if first is None:
                first = key
            cur.append(node)
            cur_bytes += need
        pg = self.alloc()
        self._emit(pg, P_LEAF, cur)
        self.stats['leaf'] += 1
        level.append((first, pg))
        depth = 1
        while len(level) > 1:
            nxt, cur, cur_bytes, first = [], [], 0, None
            for key, child in level:
                k = b'' if not cur else key        # the first key of a branch page is implicit
                node = struct.pack('<HHHH', child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, len(k)) + k
                need = ((len(node) + 1) & ~1) + 2
                if cur and cur_bytes + need > self.psize - PAGEHDRSZ:
                    pg = self.alloc()
                    self._emit(pg, P_BRANCH, cur)
                    self.stats['branch'] += 1
                    nxt.append((first, pg))
                    cur, cur_bytes, first = [], 0, None
                    node = struct.pack('<HHHH', child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, 0)
                    need = ((len(node) + 1) & ~1) + 2
                if first is None:
                    first = key
                cur.append(node)
                cur_bytes += need
            pg = self.alloc()
            self._emit(pg, P_BRANCH, cur)
            self.stats['branch'] += 1
            nxt.append((first, pg))
            level = nxt
            depth += 1
        return (depth, self.stats['branch'] - b0, self.stats['leaf'] - l0, self.stats['overflow'] - o0, len(records),
                level[0][1])


def write_environment(path, databases, psize=4096, subdir=True):
    """Create `path/data.mdb` holding `databases`: {name bytes or None: {key bytes: value bytes}}; None = records of
    the unnamed MAIN database (cannot be mixed with named ones that collide with its keys)."""
    w = _PageWriter(psize)
    main_records, main_flags = [], []
    for name, recs in databases.items():
        if name is None:
            for k, v in recs.items():
                main_records.append((bytes(k), bytes(v), 0))
    for name, recs in databases.items():
        if name is None:
            continue
        depth, br, lf, ov, n, root = w.build_tree(sorted((bytes(k), bytes(v)) for k, v in recs.items()))
        main_records.append((bytes(name), _DB.pack(0, 0, depth, br, lf, ov, n, root), F_SUBDATA))
    main_records.sort(key=lambda r: r[0])
    depth, br, lf, ov, n, root = w.build_tree([(k, v) for k, v, _ in main_records], [f for _, _, f in main_records])
    last = w.next - 1
    fname = os.path.join(path, 'data.mdb') if subdir else path
    if subdir:
        os.makedirs(path, exist_ok=True)
    with builtins.open(fname, 'wb') as f:
        for i in range(2):
            buf = bytearray(psize)
            struct.pack_into('<QHHHH', buf, 0, i, 0, P_META, 0, 0)
            _META.pack_into(buf, PAGEHDRSZ, MAGIC, 1, 0, (last + 1) * psize)
            _DB.pack_into(buf, PAGEHDRSZ + 24, psize, 0, 0, 0, 0, 0, 0, INVALID)            # FREE db (empty)
            _DB.pack_into(buf, PAGEHDRSZ + 24 + 48, 0, 0, depth, br, lf, ov, n, root)      # MAIN db
            struct.pack_into('<QQ', buf, PAGEHDRSZ + 24 + 96, last, 1 if i == 1 else 0)   # newer txn on page 1
            f.write(buf)
        pg = 2
        for pgno in sorted(w.pages):
            assert pgno == pg, (pgno, pg)
            f.write(w.pages[pgno])
            pg += len(w.pages[pgno]) // psize
    return fname


# ----------------------------------------------------------------------------------------------------------------
# the reference's use of it (data/lmdb_dataset.py:59-88)
# ----------------------------------------------------------------------------------------------------------------
class LMDBImageStore(object):
    """Host-side mirror of `LMDBDataset.prepare / search_image / default_unpack`: one environment per path, the named
    database `image`, `get(data_id)` -> encoded image bytes, decoded to **uint8 BGR HWC** (cv2.imdecode(...,
    IMREAD_COLOR)'s channel order; the reference converts to float32 on the host, here `DevicePipeline` does it on
    the GPU)."""

    def __init__(self, lmdb_paths, db_name=b'image'):
        if isinstance(lmdb_paths, str):
            lmdb_paths = [lmdb_paths]
        self.envs, self.txns = [], {}
        for path in lmdb_paths:
            path = os.path.join(path, '')
            env = open(path, max_dbs=1, lock=False)
            self.envs.append(env)
            self.txns[path] = env.begin(db=env.open_db(db_name))

    def search_image(self, data_id, path):
        if isinstance(data_id, str):
            data_id = data_id.encode()
        maybe_image = self.txns[os.path.join(path, '')].get(data_id)
        assert maybe_image is not None, 'image %s not found at %s' % (data_id, path)
        return maybe_image

    def default_unpack(self, data_id, meta):
        import io

        import numpy as np
        from PIL import Image
        data = self.search_image(data_id, meta['db_path'])
        rgb = np.array(Image.open(io.BytesIO(data)).convert('RGB'))
        meta['image'] = np.ascontiguousarray(rgb[:, :, ::-1])
        return meta

    def close(self):
        for env in self.envs:
            env.close()
