"""Independent pure-Python walk of a Lance v2.0 file (footer, offset tables, protobuf wire format) -- test
infrastructure: cross-checks the native reader/writer in lance_amd/csrc/lance_file.cpp byte for byte.
Layout: /root/reference/protos/file2.proto:31-100."""
import struct


def varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return r, i


def fields(b):
    """-> [(field number, wire type, value)]; value is an int for varints, bytes otherwise"""
    i, out = 0, []
    while i < len(b):
        key, i = varint(b, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(b, i)
        elif wt == 1:
            v, i = b[i:i + 8], i + 8
        elif wt == 2:
            n, i = varint(b, i)
            v, i = b[i:i + n], i + n
        elif wt == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(f"wire type {wt}")
        out.append((fn, wt, v))
    return out


def packed(v):
    if isinstance(v, int):
        return [v]
    i, out = 0, []
    while i < len(v):
        x, i = varint(v, i)
        out.append(x)
    return out


class Probe:
    def __init__(self, path):
        self.b = b = open(path, "rb").read()
        self.cm_start, self.cmo, self.gbo, self.ngb, self.ncol, self.major, self.minor = struct.unpack("<QQQIIHH", b[-40:-4])
        self.magic = b[-4:]
        self.global_buffers = [struct.unpack_from("<QQ", b, self.gbo + 16 * g) for g in range(self.ngb)]
        self.column_meta = []          # raw ColumnMetadata bytes per column
        self.pages = []                # per column: [(offsets, sizes, length, encoding bytes, priority)]
        for c in range(self.ncol):
            pos, sz = struct.unpack_from("<QQ", b, self.cmo + 16 * c)
            raw = b[pos:pos + sz]
            self.column_meta.append(raw)
            pages = []
            for fn, wt, v in fields(raw):
                if fn == 2:
                    pg = {"offsets": [], "sizes": [], "length": 0, "encoding": b"", "priority": 0}
                    for f2, w2, v2 in fields(v):
                        if f2 == 1:
                            pg["offsets"] += packed(v2)
                        elif f2 == 2:
                            pg["sizes"] += packed(v2)
                        elif f2 == 3:
                            pg["length"] = v2
                        elif f2 == 4:
                            pg["encoding"] = v2
                        elif f2 == 5:
                            pg["priority"] = v2
                    pages.append(pg)
            self.pages.append(pages)
        # file descriptor
        pos, sz = self.global_buffers[0]
        self.schema_fields, self.metadata, self.length = [], {}, 0
        for fn, wt, v in fields(b[pos:pos + sz]):
            if fn == 2:
                self.length = v
            elif fn == 1:
                for f2, w2, v2 in fields(v):
                    if f2 == 1:
                        self.schema_fields.append({a: c for a, _, c in fields(v2)})
                    elif f2 == 5:
                        kv = {a: c for a, _, c in fields(v2)}
                        self.metadata[kv.get(1, b"").decode()] = kv.get(2, b"")

    def global_buffer(self, i):
        pos, sz = self.global_buffers[i]
        return self.b[pos:pos + sz]

    def page_bytes(self, col, page=0, buf=0):
        pg = self.pages[col][page]
        return self.b[pg["offsets"][buf]:pg["offsets"][buf] + pg["sizes"][buf]]
