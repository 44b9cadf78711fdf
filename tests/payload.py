"""Independent (pure Python / numpy) reader of the container and of the serialized Huffman
record, written from the format description only (reference Huffman.hpp:163-239,
Decode.hpp:52-106, GPUPipelines.hpp:189-193): used by the tests to check what the library
writes without going through the library's own decoder."""
import struct

import numpy as np


def _align(off, a):
    return (off + a - 1) // a * a


def parse_huffman_record(b):
    """dict with primary_count, dict_size, chunk_size, bits[], entry[], first[64], entry_tab[64],
    keys[dict], units[], outlier_idx[], outliers[]."""
    off = 0
    primary, = struct.unpack_from("<Q", b, off); off += 8
    dict_size, chunk = struct.unpack_from("<ii", b, off); off += 8
    off = _align(off, 8)
    hm, = struct.unpack_from("<Q", b, off); off += 8
    meta = np.frombuffer(b, dtype="<u8", count=hm, offset=off); off += 8 * hm
    dbs, = struct.unpack_from("<Q", b, off); off += 8
    assert dbs == 8 * 128 + 8 * dict_size
    first = np.frombuffer(b, dtype="<u8", count=64, offset=off)
    entry_tab = np.frombuffer(b, dtype="<u8", count=64, offset=off + 512)
    keys = np.frombuffer(b, dtype="<u8", count=dict_size, offset=off + 1024)
    off += dbs
    off = _align(off, 8)
    nunits, = struct.unpack_from("<Q", b, off); off += 8
    off = _align(off, 8)
    units = np.frombuffer(b, dtype="<u8", count=nunits, offset=off); off += 8 * nunits
    oc, = struct.unpack_from("<Q", b, off); off += 8
    oidx = np.frombuffer(b, dtype="<u8", count=oc, offset=off); off += 8 * oc
    oval = np.frombuffer(b, dtype="<i8", count=oc, offset=off); off += 8 * oc
    nchunk = hm // 2
    # behind the reference's payload, optional: the decoder's synchronisation points
    # ([u64 "MGHSYNC1"][u32 x 64 per chunk], highlevel.hip: PayloadLayout)
    sync = None
    if off != len(b):
        assert len(b) - off == 8 + 256 * nchunk, (off, len(b), nchunk)
        tag, = struct.unpack_from("<Q", b, off)
        assert tag == SYNC_TAG, hex(tag)
        sync = np.frombuffer(b, dtype="<u4", count=64 * nchunk, offset=off + 8).reshape(nchunk, 64)
        off += 8 + 256 * nchunk
    assert off == len(b), (off, len(b))
    return dict(primary_count=primary, dict_size=dict_size, chunk_size=chunk, bits=meta[:nchunk],
                entry=meta[nchunk:], first=first, entry_tab=entry_tab, keys=keys, units=units,
                outlier_idx=oidx, outliers=oval, sync=sync)


SYNC_TAG = int.from_bytes(b"MGHSYNC1", "little")


def expected_sync_points(rec, symbols):
    """What the synchronisation points of a record must be, from the symbols and the decodebook alone:
    entry k of a chunk = (distance of the first code that starts at or behind bit k * B from that bit)
    << 16 | index of its symbol in the chunk, B = ceil(bits of the chunk / 64); (0, symbols in the
    chunk) where no code starts behind k * B; entry 0 = 0."""
    n, chunk = int(rec["primary_count"]), int(rec["chunk_size"])
    length = {}
    first = [int(x) for x in rec["first"]]
    entry = [int(x) for x in rec["entry_tab"]] + [int(rec["dict_size"])]
    nxt = int(rec["dict_size"])
    for l in range(63, 0, -1):            # keys[entry[l] .. next used entry) carry codes of length l
        if first[l] != 2 ** 64 - 1:
            for k in range(entry[l], nxt):
                length[int(rec["keys"][k])] = l
            nxt = entry[l]
    out = np.zeros((len(rec["bits"]), 64), dtype=np.uint32)
    for c in range(len(rec["bits"])):
        sy = symbols[c * chunk:min(n, (c + 1) * chunk)]
        starts = np.concatenate([[0], np.cumsum([length[int(x)] for x in sy])]).astype(np.int64)
        total = int(starts[-1])
        assert total == int(rec["bits"][c])
        B = (total + 63) // 64
        for k in range(1, 64):
            i = int(np.searchsorted(starts[:-1], k * B, side="left")) if B else len(sy)
            out[c, k] = ((int(starts[i]) - k * B) << 16 | i) if i < len(sy) else len(sy)
    return out


def decode_huffman_record(rec, max_chunks=None):
    """Every chunk through the decoder restated from Decode.hpp:52-106 (oracle/huffman_ref.py;
    bit-serial, slow: for small inputs). Returns the symbols as int64."""
    from oracle import huffman_ref
    first = [int(x) for x in rec["first"]]
    entry = [int(x) for x in rec["entry_tab"]]
    keys = [int(x) for x in rec["keys"]]
    n, chunk = rec["primary_count"], rec["chunk_size"]
    out = np.zeros(n, dtype=np.int64)
    nchunk = len(rec["bits"]) if max_chunks is None else min(max_chunks, len(rec["bits"]))
    for c in range(nchunk):
        total = int(rec["bits"][c])
        base = int(rec["entry"][c])          # dH_meta[n_chunk + chunk_id]: first code unit of the chunk
        want = min(n, (c + 1) * chunk) - c * chunk
        sym = huffman_ref.decode(rec["units"][base:base + (total + 63) // 64], total, first, entry, keys)
        assert len(sym) == want, (c, len(sym), want)
        out[c * chunk:c * chunk + want] = sym
    return out


def write_huffman_record(symbols, dict_size, chunk, codebook, first, entry, keys,
                         outlier_idx=(), outliers=()):
    """The serialized record of Huffman.hpp:163-239 written from a codebook given as the reference
    holds it (codebook[s] = (length << 56) | codeword, first[64] / entry[64] / keys[dict]; None
    entries -- never written by GenerateCW -- become 0 like the reference's zero-filled workspace):
    what the stock ENCODER would hand to a decoder. Small inputs only."""
    symbols = [int(x) for x in symbols]
    n = len(symbols)
    nchunk = (n - 1) // chunk + 1
    bits_per_chunk, units = [], []
    for c in range(nchunk):
        word, nb, out = 0, 0, []
        total = 0
        for sy in symbols[c * chunk:(c + 1) * chunk]:
            cw = int(codebook[sy])
            l, v = cw >> 56, cw & ((1 << 56) - 1)
            total += l
            for k in range(l):                      # MSB first into H = u64 units (Deflate.hpp:33-76)
                word = (word << 1) | ((v >> (l - 1 - k)) & 1)
                nb += 1
                if nb == 64:
                    out.append(word)
                    word, nb = 0, 0
        if nb:
            out.append(word << (64 - nb))
        bits_per_chunk.append(total)
        units.append(out)
    offs, acc = [], 0
    for u in units:
        offs.append(acc)
        acc += len(u)
    b = bytearray()
    b += struct.pack("<Q", n) + struct.pack("<ii", dict_size, chunk)
    b += struct.pack("<Q", 2 * nchunk)
    b += np.asarray(bits_per_chunk + offs, dtype="<u8").tobytes()
    b += struct.pack("<Q", 8 * 128 + 8 * dict_size)
    b += np.asarray([0 if x is None else int(x) for x in first], dtype="<u8").tobytes()
    b += np.asarray([0 if x is None else int(x) for x in entry], dtype="<u8").tobytes()
    b += np.asarray([int(x) for x in keys], dtype="<u8").tobytes()
    b += struct.pack("<Q", acc)
    b += np.asarray([w for u in units for w in u], dtype="<u8").tobytes()
    b += struct.pack("<Q", len(outlier_idx))
    b += np.asarray(outlier_idx, dtype="<u8").tobytes() + np.asarray(outliers, dtype="<i8").tobytes()
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def split_container(buf, metadata_size):
    """[(size, payload bytes)] of the subdomain records behind the header."""
    b = bytes(buf)
    off, out = metadata_size, []
    while off < len(b):
        size, = struct.unpack_from("<Q", b, off)
        off += 8
        out.append(b[off:off + size])
        off += size
    assert off == len(b)
    return out
