"""TEST INFRASTRUCTURE ONLY -- never imported by the product (mgard_amd/), only by tests/.

CPU restatement (numpy / plain Python, small inputs) of the pieces of the reference's MGARD-X
Huffman stage that decide whether a payload written by this repository is what the stock decoder
expects, and whether a payload the stock encoder writes is read by this repository's decoder:

  * generate_cw()   -- canonical codewords + the first[] / entry[] tables from sorted code lengths
                       (include/mgard-x/Lossless/ParallelHuffman/GenerateCW.hpp:40-232, the ten
                       "Operations" of GenerateCWFunctor executed sequentially)
  * get_codebook()  -- the data flow around it: stable sort of the symbols by frequency, the cut at
                       the first non-zero frequency, GenerateCW, the two array reversals and the
                       reordering by symbol (GetCodebook.hpp:23-147)
  * decode()        -- the bit-serial canonical decoder (Decode.hpp:52-106)
  * generate_cl()   -- the parallel code-LENGTH construction (GenerateCL.hpp:58-690), sequentially.
                       The reference reads histogram[lNodesCur + curLeavesNum] (:331-336), one
                       element past the array whenever every remaining leaf joins a merge (4 of
                       10 random histograms): its result is then not a function of the histogram
                       and generate_cl raises ReferenceReadsOutOfBounds. Where it is defined it is
                       optimal and monotone (asserted), and equals the library's two-queue
                       construction except for how some ties are broken (5 of 153 random cases);
                       the decodebook travels with the payload (Huffman.hpp:163-239), so either
                       side decodes the other's records (tests/test_huffman_reference_rules.py).

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


class ReferenceReadsOutOfBounds(Exception):
    """GenerateCL.hpp:331-336 reads histogram[lNodesCur + curLeavesNum]; when every remaining leaf
    takes part in the merge that index is one past the end of the array (undefined in the
    reference, so there is nothing to restate for such an input)."""


UINT_MAX = 0xFFFFFFFF


def generate_cl(freq_ascending):
    """GenerateCLFunctor (GenerateCL.hpp:58-690), the Operations executed sequentially: the
    two-phase parallel Huffman code-LENGTH construction (select the two smallest nodes, then meld
    every leaf not heavier than their sum with the queued internal nodes pairwise). Input: the
    non-zero frequencies in ASCENDING order (GetCodebook.hpp:70-88 passes the same array as
    `histogram` and `lNodesFreq`). Returns CL[i] per sorted position.

    Unpinned: no reference-held vector exists for this stage; the tests only compare it with the
    library's own construction and say where the two differ."""
    f = [int(x) for x in freq_ascending]
    n = len(f)
    MOD = lambda a, b: ((a % b) + b) % b                     # GenerateCL.hpp:14-16
    CL = [0] * n                                             # Operation1 (:58-81)
    lLeader = [-1] * n
    iFreq = [0] * n
    iLeader = [-1] * n
    front = rear = cur = size = 0
    while cur < n or size > 1:                               # LoopCondition1 (:83-93)
        # ---- Operation2 (:95-296): the two least frequent of {2 leaves, 2 internal nodes}
        mid = [[UINT_MAX, 0] for _ in range(4)]
        if cur < n:
            mid[0] = [f[cur], 1]
        if cur < n - 1:
            mid[1] = [f[cur + 1], 1]
        if size >= 1:
            mid[2] = [iFreq[front], 0]
        if size >= 2:
            mid[3] = [iFreq[MOD(front + 1, n)], 0]
        for a, b in ((1, 3), (0, 2), (0, 1), (2, 3), (1, 2)):  # the sorting network (:141-182)
            if mid[a][0] > mid[b][0]:
                mid[a], mid[b] = mid[b], mid[a]
        minFreq = mid[0][0]
        if mid[1][0] < UINT_MAX:                             # only one node left: no merge
            minFreq += mid[1][0]
        iFreq[rear] = minFreq
        iLeader[rear] = -1
        for k in (0, 1):
            if mid[k][0] < UINT_MAX:
                if mid[k][1]:
                    lLeader[cur] = rear
                    CL[cur] += 1
                    cur += 1
                else:
                    iLeader[front] = rear
                    front = MOD(front + 1, n)
        size = MOD(rear - front, n)
        # ---- Operation3/4 (:298-330): leaves not heavier than the node just made
        curLeavesNum = 0
        for i in range(cur, n):
            if f[i] <= minFreq:
                curLeavesNum = max(curLeavesNum, i - cur + 1)
        # ---- Operation5 (:332-403)
        mergeRear, mergeFront = rear, front
        if (curLeavesNum + size) % 2 == 0:
            front = rear
        else:
            cond = False
            if size != 0:
                if curLeavesNum == 0:
                    cond = True
                else:
                    if cur + curLeavesNum >= n:
                        raise ReferenceReadsOutOfBounds()
                    cond = f[cur + curLeavesNum] <= iFreq[MOD(rear - 1, n)]
            if cond:
                mergeRear = MOD(mergeRear - 1, n)
                front = MOD(rear - 1, n)
            else:
                front = rear
                curLeavesNum -= 1
        copy_idx = list(range(cur, cur + curLeavesNum))      # Operation4's copy (before cur moves)
        cur += curLeavesNum
        rear = MOD(rear + 1, n)
        tempLength = curLeavesNum + MOD(mergeRear - mergeFront, n)
        if tempLength > 0:                                   # BranchCondition1 (:405-408)
            # ---- Operations 6-11 (:410-606): merge path = stable merge, leaf first on ties
            temp = []
            a, b = 0, mergeFront
            while a < len(copy_idx) and MOD(mergeRear - b, n) > 0:
                if f[copy_idx[a]] <= iFreq[b]:
                    temp.append((f[copy_idx[a]], copy_idx[a], 1))
                    a += 1
                else:
                    temp.append((iFreq[b], b, 0))
                    b = MOD(b + 1, n)
            while a < len(copy_idx):
                temp.append((f[copy_idx[a]], copy_idx[a], 1))
                a += 1
            while MOD(mergeRear - b, n) > 0:
                temp.append((iFreq[b], b, 0))
                b = MOD(b + 1, n)
            # ---- Operation12 (:608-632): meld pairwise into new internal nodes
            for i in range(tempLength // 2):
                ind = MOD(rear + i, n)
                iFreq[ind] = temp[2 * i][0] + temp[2 * i + 1][0]
                iLeader[ind] = -1
                for fr, idx, leaf in (temp[2 * i], temp[2 * i + 1]):
                    if leaf:
                        lLeader[idx] = ind
                        CL[idx] += 1
                    else:
                        iLeader[idx] = ind
            rear = MOD(rear + tempLength // 2, n)            # Operation13 (:634-642)
        # ---- Operation14 (:644-658): leaves follow their leader one step up
        for i in range(n):
            if lLeader[i] != -1 and iLeader[lLeader[i]] != -1:
                lLeader[i] = iLeader[lLeader[i]]
                CL[i] += 1
        size = MOD(rear - front, n)                          # Operation15 (:660-668)
    return CL
