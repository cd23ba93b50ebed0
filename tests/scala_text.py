"""Pure-Python, statement-by-statement transliteration of two routines of the reference's Scala text, used
as an independent second opinion on the C oracle for SMALL cases (slow by design).

  sw_extend : SWUtil.SWExtend, src/main/scala/cs/ucla/edu/bwaspark/util/SWUtil.scala:61-230
  sw_align  : SWUtil.SWAlign,  SWUtil.scala:417-570   (sw_align2: 583-601)

Written from the Scala source independently of oracle/bpsw_oracle.c (different data layout: a list of
[h, e] pairs like the Scala EHType objects), so a transcription slip in one is unlikely to be mirrored.
"""
MINUS_INF = -0x40000000
KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x20000, 0x40000, 0x80000


def sw_extend(query, target, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0, m=5):
    q_len, t_len = len(query), len(target)
    eh = [[0, 0] for _ in range(q_len + 1)]                       # [h, e]
    qp = [mat[k * m + query[j]] for k in range(m) for j in range(q_len)]
    oe_del, oe_ins = o_del + e_del, o_ins + e_ins
    eh[0][0] = h0
    eh[1][0] = h0 - oe_ins if h0 > oe_ins else 0
    j = 2
    while j <= q_len and eh[j - 1][0] > e_ins:
        eh[j][0] = eh[j - 1][0] - e_ins
        j += 1
    mx = max(mat)
    max_ins = int((q_len * mx + end_bonus - o_ins) / float(e_ins) + 1.0)
    max_ins = max(max_ins, 1)
    w = min(w, max_ins)
    max_del = int((q_len * mx + end_bonus - o_del) / float(e_del) + 1.0)
    max_del = max(max_del, 1)
    w = min(w, max_del)
    mx = h0
    max_i = max_j = max_ie = gscore = -1
    max_off = 0
    beg, end = 0, q_len
    brk = False
    i = 0
    while i < t_len and not brk:
        f = 0
        mm = 0
        mj = -1
        qptr = target[i] * q_len
        h1 = h0 - (o_del + e_del * (i + 1))
        if h1 < 0:
            h1 = 0
        if beg < i - w:
            beg = i - w
        if end > i + w + 1:
            end = i + w + 1
        if end > q_len:
            end = q_len
        j = beg
        while j < end:
            h, e = eh[j]
            eh[j][0] = h1
            h += qp[qptr + j]
            if h < e:
                h = e
            if h < f:
                h = f
            h1 = h
            if mm <= h:
                mj = j
                mm = h
            t = h - oe_del
            if t < 0:
                t = 0
            e -= e_del
            if e < t:
                e = t
            eh[j][1] = e
            t = h - oe_ins
            if t < 0:
                t = 0
            f -= e_ins
            if f < t:
                f = t
            j += 1
        eh[end][0] = h1
        eh[end][1] = 0
        if j == q_len:
            if gscore <= h1:
                max_ie = i
                gscore = h1
        if mm == 0:
            brk = True
        else:
            if mm > mx:
                mx = mm
                max_i = i
                max_j = mj
                if max_off < abs(mj - i):
                    max_off = abs(mj - i)
            elif zdrop > 0:
                # Scala: `if (A) if (B) isBreak = true else if (C) isBreak = true` -- the else binds to the inner if
                if (i - max_i) > (mj - max_j):
                    if mx - mm - ((i - max_i) - (mj - max_j)) * e_del > zdrop:
                        brk = True
                    elif mx - mm - ((mj - max_j) - (i - max_i)) * e_ins > zdrop:
                        brk = True
            if not brk:
                j = mj
                while j >= beg and eh[j][0] > 0:
                    j -= 1
                beg = j + 1
                j = mj + 2
                while j <= end and eh[j][0] > 0:
                    j += 1
                end = j
        i += 1
    return [mx, max_j + 1, max_i + 1, max_ie + 1, gscore, max_off]


def sw_align(query, target, mat, a, b, o_del, e_del, o_ins, e_ins, xtra, m=5):
    q_len, t_len = len(query), len(target)
    max_score = 255 - abs(b)
    oe_del, oe_ins = o_del + e_del, o_ins + e_ins
    eh = [[0, 0] for _ in range(q_len)]
    qp = [mat[k * m + query[j]] for k in range(m) for j in range(q_len)]
    best, tend = [], []
    min_score = (xtra & 0xffff) if (xtra & KSW_XSUBO) else 0x10000
    end_score = (xtra & 0xffff) if (xtra & KSW_XSTOP) else 0x10000
    mx, max_i, max_j = MINUS_INF, -1, -1
    i = 0
    brk = False
    while i < t_len and not brk:
        f = h1 = mm = 0
        mj = -1
        qptr = target[i] * q_len
        for j in range(q_len):
            h, e = eh[j]
            eh[j][0] = h1
            h += qp[qptr + j]
            if h < e:
                h = e
            if h < f:
                h = f
            h1 = h
            if mm < h:
                mj = j
                mm = h
            t = h - oe_del
            if t < 0:
                t = 0
            e -= e_del
            if e < t:
                e = t
            eh[j][1] = e
            t = h - oe_ins
            if t < 0:
                t = 0
            f -= e_ins
            if f < t:
                f = t
        if mm >= min_score:
            if not best or tend[-1] + 1 != i:
                best.append(mm)
                tend.append(i)
            elif best[-1] < mm:
                best[-1] = mm
                tend[-1] = i
        if mm > mx:
            mx, max_i, max_j = mm, i, mj
            if mx >= end_score or mx >= max_score:
                brk = True
        i += 1
    if mx >= max_score:
        mx = 255
    out = [mx, max_i, -1, -1, -1, -1, -1]
    if mx != 255:
        out[2] = max_j
        if best:
            tmp = (mx + a - 1) // a
            low, high = max_i - tmp, max_i + tmp
            for k in range(len(best)):
                if (tend[k] < low or tend[k] > high) and best[k] > out[3]:
                    out[3], out[4] = best[k], tend[k]
    return out


def sw_align2(query, target, mat, a, b, o_del, e_del, o_ins, e_ins, xtra):
    aln = sw_align(query, target, mat, a, b, o_del, e_del, o_ins, e_ins, xtra)
    if (xtra & KSW_XSTART) == 0 or ((xtra & KSW_XSUBO) and aln[0] < (xtra & 0xffff)):
        return aln
    q, t = list(query), list(target)
    q[:aln[2] + 1] = q[:aln[2] + 1][::-1]
    t[:aln[1] + 1] = t[:aln[1] + 1][::-1]
    rev = sw_align(q[:aln[2] + 1], t, mat, a, b, o_del, e_del, o_ins, e_ins, KSW_XSTOP | aln[0])
    if aln[0] == rev[0]:
        aln[5] = aln[1] - rev[1]
        aln[6] = aln[2] - rev[2]
    return aln
