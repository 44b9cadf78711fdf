"""TEST INFRASTRUCTURE ONLY -- never imported by the product (mgard_amd/), only by tests/.

CPU restatement (numpy / plain Python, small inputs) of the three pieces of the reference's
MGARD-X Huffman stage that decide whether a payload written by this repository is what the stock
decoder expects:

  * generate_cw()   -- canonical codewords + the first[] / entry[] tables from sorted code lengths
                       (include/mgard-x/Lossless/ParallelHuffman/GenerateCW.hpp:40-232, the ten
                       "Operations" of GenerateCWFunctor executed sequentially)
  * get_codebook()  -- the data flow around it: stable sort of the symbols by frequency, the cut at
                       the first non-zero frequency, GenerateCW, the two array reversals and the
                       reordering by symbol (GetCodebook.hpp:23-147)
  * decode()        -- the bit-serial canonical decoder (Decode.hpp:52-106)

What is NOT restated: GenerateCL.hpp (the parallel code-LENGTH construction). Code lengths are an
input here; the tests take them from the library's own codebook and check the two properties
GenerateCW relies on (lengths optimal; non-increasing with frequency along the sorted order).
Among several optimal length assignments the reference's GenerateCL may pick another one than
this repository's two-queue construction; the decodebook travels with the payload
(Huffman.hpp:163-239), so either is decodable by the other side.

Pin status: no golden vectors exist for this stage (the reference's tests round-trip the legacy
CPU Huffman only: tests/src/test_compressors.cpp:15-39); this file is pinned by being a
line-by-line readable restatement of the cited reference lines and nothing else.
"""
import numpy as np

TYPE_BW = 64                      # sizeof(H) * 8 for H = unsigned long long (Huffman.hpp: H = uint64)
H_MAX = (1 << 64) - 1             # std::numeric_limits<H>::max()
MASK = H_MAX


def generate_cw(cl_sorted_ascending_freq):
    """GenerateCWFunctor (GenerateCW.hpp:40-232). Input: CL[i] = code length of the i-th symbol in
    ASCENDING frequency order (what GenerateCL leaves in workspace.CL_subarray), all non-zero.
    Returns (CW, first, entry): CW[i] in the same (ascending-frequency) order, encoded as
    (length << 56) | flipped codeword (Operation9), first/entry as the decoder reads them; entries
    the functor never writes are None."""
    CL = [int(x) for x in cl_sorted_ascending_freq]
    n = len(CL)
    CW = [0] * n
    first = [None] * TYPE_BW
    entry = [None] * TYPE_BW
    # Operation1 (:40-52): reverse CL in place -> descending frequency, lengths ascending
    CL.reverse()
    # Operation2 (:54-71)
    CCL = CL[0]
    CDPI = 0
    newCDPI = n - 1
    entry[CCL] = 0
    CW[CDPI] = 0                                         # edge case: only one input symbol
    first[CCL] = CW[CDPI] ^ ((1 << CL[CDPI]) - 1)
    entry[CCL + 1] = 1
    # Operation3 (:73-84): unused short lengths are skipped by the decoder
    for i in range(CCL):
        first[i] = H_MAX
        entry[i] = 0
    # LoopCondition1 (:86-92)
    while CDPI < n - 1:
        # Operation4 (:94-103): last index of the current length
        for i in range(n - 1):
            if CL[i + 1] > CCL:
                newCDPI = min(newCDPI, i)
        # Operation5 (:105-129)
        updateEnd = TYPE_BW if newCDPI >= n - 1 else CL[newCDPI + 1]
        curEntryVal = entry[CCL]
        numCCL = newCDPI - CDPI + 1
        if CDPI == 0:
            CW[newCDPI] = 0
        else:
            CW[newCDPI] = CW[CDPI]                       # pre-stored by Operation8
        # Operation6 (:131-155): the group's codewords count down along the index
        base = CW[newCDPI]
        for i in range(CDPI, newCDPI):
            CW[i] = base + (newCDPI - i)
        for i in range(CCL + 1, updateEnd):
            entry[i] = curEntryVal + numCCL
        if updateEnd < TYPE_BW:
            entry[updateEnd] = curEntryVal + numCCL
        # Operation7 (:157-173): flip the least significant CL bits of the group's largest codeword
        first[CCL] = CW[CDPI] ^ ((1 << CL[CDPI]) - 1)
        for i in range(CCL + 1, updateEnd):
            first[i] = H_MAX
        # Operation8 (:175-207): add and shift -- next canonical code
        if newCDPI < n - 1:
            CLDiff = CL[newCDPI + 1] - CL[newCDPI]
            CW[newCDPI + 1] = ((CW[CDPI] + 1) << CLDiff) & MASK
            CCL = CL[newCDPI + 1]
            newCDPI += 1
        CDPI = newCDPI
        newCDPI = n - 1
    # Operation9 (:209-222): length into the highest 8 bits, codeword bits flipped
    for i in range(n):
        CW[i] = ((CW[i] | ((CL[i] & 0xFF) << (TYPE_BW - 8))) ^ ((1 << CL[i]) - 1)) & MASK
    # Operation10 (:224-236): reverse the (partial) codebook back to ascending frequency
    CW.reverse()
    return CW, first, entry


def get_codebook(freq, length_of_symbol):
    """GetCodebook (GetCodebook.hpp:23-147) with the code lengths supplied by the caller
    (length_of_symbol[s], 0 for unused symbols) in place of GenerateCLKernel (:76-88).
    Returns (codebook[dict], first[64], entry[64], keys[dict])."""
    freq = np.asarray(freq, dtype=np.uint64)
    dict_size = len(freq)
    # :44-57 qcode = 0..dict-1, SortByKey(freq, qcode) ascending; radix / std::stable_sort: stable
    qcode = np.argsort(freq, kind="stable")
    sfreq = freq[qcode]
    # :59-67 first non-zero frequency
    nzi = int(np.searchsorted(sfreq, 1, side="left"))
    nz = dict_size - nzi
    codebook_sorted = [0] * dict_size
    first = [None] * TYPE_BW
    entry = [None] * TYPE_BW
    if nz > 0:
        CL = [int(length_of_symbol[int(s)]) for s in qcode[nzi:]]
        CW, first, entry = generate_cw(CL)               # :119-123 on the non-zero tail
        codebook_sorted[nzi:] = CW
    # :125-128 ReverseArray on the whole codebook and on qcode: descending frequency first
    codebook_sorted.reverse()
    keys = [int(s) for s in qcode[::-1]]
    # :130-138 ReorderByIndex: codebook[qcode[i]] = codebook_sorted[i]
    codebook = [0] * dict_size
    for i, s in enumerate(keys):
        codebook[s] = codebook_sorted[i]
    return codebook, first, entry, keys


def decode(units, total_bits, first, entry, keys, max_symbols=None):
    """DecodeFunctor::Operation2 for ONE chunk (Decode.hpp:52-106): `units` are the H = u64 code
    units of the chunk (bits consumed MSB first), total_bits = dH_meta[chunk]. Returns the list of
    decoded symbols."""
    def bit(i):
        return (int(units[i // TYPE_BW]) >> (TYPE_BW - 1 - i % TYPE_BW)) & 1
    out = []
    v = bit(0)                                           # :64-65 the first bit
    l = 1
    i = 0
    while i < total_bits:                                # :68
        while v < first[l]:                              # :69-77 append the next bit
            i += 1
            v = (v << 1) | bit(i)
            l += 1
        out.append(keys[entry[l] + v - first[l]])        # :103
        if max_symbols is not None and len(out) >= max_symbols:
            break
        i += 1                                           # :105-113 start the next codeword
        if i < total_bits:
            v = bit(i)
        l = 1
    return out
