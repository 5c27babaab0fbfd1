// k2_tie.hip -- origin pointer of periodic blocks (block == w^k, k >= 2), reference-exact.
//
// The BWT bytes of a block are unique, but when the block is a k-th power the k rotations
// equal to rotation 0 are indistinguishable and the reference's origin pointer is "wherever
// its sorter left suffix 0 inside that class": a function of the sorter's internal history
// (pivot swaps, merge order, budget exhaustion, the Larsson-Sadakane give-up).
// There is no closed form (DESIGN.md "orig on periodic blocks" tabulates it), so for exactly
// these blocks one lane per block runs the reference's sorter itself, statement by statement,
// on the device, and only its origin pointer is kept.  Aperiodic blocks never come here.
//
// Follows /root/reference/Sources/bzip2JavApi/BZip2DivSufSort.swift ("D:" below), reading the
// Java-original statements kept in its comments where the Swift text departs from them
// (SURVEY.md §2.3 D12-D21), plus BZip2TRBudget.swift, BZip2StackEntry.swift,
// BZip2PartitionResult.swift.  PA/ISA/ISAd/buf are offsets into SA exactly as in the reference,
// so reads outside the logical sub-arrays see the same memory the reference would see.
#include "ctx.hpp"

namespace {

#define STACK_SIZE 64
#define BUCKET_A_SIZE 256
#define BUCKET_B_SIZE 65536
#define SS_BLOCKSIZE 1024
#define INSERTIONSORT_THRESHOLD 8

struct stack_entry_t { int a, b, c, d; };   /* BZip2StackEntry.swift:29-66 */
struct tr_budget_t { int budget, chance; }; /* BZip2TRBudget.swift:29-64 */
struct dss_t { int *SA; const u8 *T; int n; int sa_len; int trap; };

/* D:58-67 */
__device__ static const int log2table[256] = {
    -1,0,1,1,2,2,2,2,3,3,3,3,3,3,3,3,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,
    6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,
    6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,6,
    7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,
    7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,
    7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,
    7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7,7};

#define SA (D->SA)
#define T(i) ((int)D->T[(i)])
#define SWAP(x, y) do { int t_ = SA[x]; SA[x] = SA[y]; SA[y] = t_; } while (0)
#define PUSH(A, B, C, E) do { stack[ssize].a = (A); stack[ssize].b = (B); stack[ssize].c = (C); stack[ssize].d = (E); ssize++; } while (0)

/* BZip2TRBudget.swift:44-54 */
static __device__ int budget_update(tr_budget_t *b, int size, int n) {
    b->budget -= n;
    if (b->budget <= 0) {
        if (--b->chance == 0) return 0;
        b->budget += size;
    }
    return 1;
}

/* ------------------------------------------------------------------ substring sort (ss*) */
/* D:99-133 */
static __device__ int ssCompare(dss_t *D, int p1, int p2, int depth) {
    int U1, U2, U1n, U2n;
    for (U1 = depth + SA[p1], U2 = depth + SA[p2], U1n = SA[p1 + 1] + 2, U2n = SA[p2 + 1] + 2;
         (U1 < U1n) && (U2 < U2n) && (T(U1) == T(U2)); ++U1, ++U2) {}
    return U1 < U1n ? (U2 < U2n ? T(U1) - T(U2) : 1) : (U2 < U2n ? -1 : 0);
}

/* D:144-211 (bzip2 patch: the last B* substring wraps around the block end) */
static __device__ int ssCompareLast(dss_t *D, int PA, int p1, int p2, int depth, int size) {
    int U1, U2, U1n, U2n;
    for (U1 = depth + SA[p1], U2 = depth + SA[p2], U1n = size, U2n = SA[p2 + 1] + 2;
         (U1 < U1n) && (U2 < U2n) && (T(U1) == T(U2)); ++U1, ++U2) {}
    if (U1 < U1n) return (U2 < U2n) ? T(U1) - T(U2) : 1;
    else if (U2 == U2n) return 1;
    for (U1 = U1 % size, U1n = SA[PA] + 2; (U1 < U1n) && (U2 < U2n) && (T(U1) == T(U2)); ++U1, ++U2) {}
    return U1 < U1n ? (U2 < U2n ? T(U1) - T(U2) : 1) : (U2 < U2n ? -1 : 0);
}

/* D:220-272 */
static __device__ void ssInsertionSort(dss_t *D, int PA, int first, int last, int depth) {
    int i, j, t, r;
    for (i = last - 2; first <= i; --i) {
        for (t = SA[i], j = i + 1; 0 < (r = ssCompare(D, PA + t, PA + SA[j], depth));) {
            do { SA[j - 1] = SA[j]; } while ((++j < last) && (SA[j] < 0));
            if (last <= j) break;
        }
        if (r == 0) SA[j] = ~SA[j];
        SA[j - 1] = t;
    }
}

/* D:281-332 */
static __device__ void ssFixdown(dss_t *D, int Td, int PA, int sa, int i, int size) {
    int j, k, v, c, d, e;
    for (v = SA[sa + i], c = T(Td + SA[PA + v]); (j = 2 * i + 1) < size; SA[sa + i] = SA[sa + k], i = k) {
        d = T(Td + SA[PA + SA[sa + (k = j++)]]);
        if (d < (e = T(Td + SA[PA + SA[sa + j]]))) { k = j; d = e; }
        if (d <= c) break;
    }
    SA[sa + i] = v;
}

/* D:340-370 */
static __device__ void ssHeapSort(dss_t *D, int Td, int PA, int sa, int size) {
    int i, m, t;
    m = size;
    if ((size % 2) == 0) {
        m--;
        if (T(Td + SA[PA + SA[sa + (m / 2)]]) < T(Td + SA[PA + SA[sa + m]])) SWAP(sa + m, sa + (m / 2));
    }
    for (i = m / 2 - 1; 0 <= i; --i) ssFixdown(D, Td, PA, sa, i, m);
    if ((size % 2) == 0) { SWAP(sa, sa + m); ssFixdown(D, Td, PA, sa, 0, m); }
    for (i = m - 1; 0 < i; --i) {
        t = SA[sa]; SA[sa] = SA[sa + i];
        ssFixdown(D, Td, PA, sa, 0, i);
        SA[sa + i] = t;
    }
}

#define SSV(v) T(Td + SA[PA + SA[(v)]])
/* D:380-404 */
static __device__ int ssMedian3(dss_t *D, int Td, int PA, int v1, int v2, int v3) {
    int T_v1 = SSV(v1), T_v2 = SSV(v2), T_v3 = SSV(v3);
    if (T_v1 > T_v2) { int t = v1; v1 = v2; v2 = t; int tt = T_v1; T_v1 = T_v2; T_v2 = tt; }
    if (T_v2 > T_v3) { if (T_v1 > T_v3) return v1; return v3; }
    return v2;
}
/* D:417-489 */
static __device__ int ssMedian5(dss_t *D, int Td, int PA, int v1, int v2, int v3, int v4, int v5) {
    int T_v1 = SSV(v1), T_v2 = SSV(v2), T_v3 = SSV(v3), T_v4 = SSV(v4), T_v5 = SSV(v5), t, tt;
    if (T_v2 > T_v3) { t = v2; v2 = v3; v3 = t; tt = T_v2; T_v2 = T_v3; T_v3 = tt; }
    if (T_v4 > T_v5) { t = v4; v4 = v5; v5 = t; tt = T_v4; T_v4 = T_v5; T_v5 = tt; }
    if (T_v2 > T_v4) { t = v2; v2 = v4; v4 = t; tt = T_v2; T_v2 = T_v4; T_v4 = tt;
                       t = v3; v3 = v5; v5 = t; tt = T_v3; T_v3 = T_v5; T_v5 = tt; }
    if (T_v1 > T_v3) { t = v1; v1 = v3; v3 = t; tt = T_v1; T_v1 = T_v3; T_v3 = tt; }
    if (T_v1 > T_v4) { t = v1; v1 = v4; v4 = t; tt = T_v1; T_v1 = T_v4; T_v4 = tt;
                       t = v3; v3 = v5; v5 = t; tt = T_v3; T_v3 = T_v5; T_v5 = tt; }
    if (T_v3 > T_v4) return v4;
    return v3;
}
/* D:499-521 */
static __device__ int ssPivot(dss_t *D, int Td, int PA, int first, int last) {
    int middle, t;
    t = last - first;
    middle = first + t / 2;
    if (t <= 512) {
        if (t <= 32) return ssMedian3(D, Td, PA, first, middle, last - 1);
        t >>= 2;
        return ssMedian5(D, Td, PA, first, first + t, middle, last - 1 - t, last - 1);
    }
    t >>= 3;
    return ssMedian3(D, Td, PA,
                     ssMedian3(D, Td, PA, first, first + t, first + (t << 1)),
                     ssMedian3(D, Td, PA, middle - t, middle, middle + t),
                     ssMedian3(D, Td, PA, last - 1 - (t << 1), last - 1 - t, last - 1));
}
/* D:528-532 */
static __device__ int ssLog(int n) {
    return ((n & 0xff00) != 0) ? 8 + log2table[(n >> 8) & 0xff] : log2table[n & 0xff];
}

/* D:542-577 */
static __device__ int ssSubstringPartition(dss_t *D, int PA, int first, int last, int depth) {
    int a, b, t;
    for (a = first - 1, b = last;;) {
        for (; (++a < b) && ((SA[PA + SA[a]] + depth) >= (SA[PA + SA[a] + 1] + 1));) SA[a] = ~SA[a];
        for (; (a < --b) && ((SA[PA + SA[b]] + depth) < (SA[PA + SA[b] + 1] + 1));) {}
        if (b <= a) break;
        t = ~SA[b]; SA[b] = SA[a]; SA[a] = t;
    }
    if (first < a) SA[first] = ~SA[first];
    return a;
}

/* D:585-883 */
static __device__ void ssMultiKeyIntroSort(dss_t *D, int PA, int first, int last, int depth) {
    stack_entry_t stack[STACK_SIZE];
    int Td, a, b, c, d, e, f, s, t, ssize, limit, v, x = 0;
    for (ssize = 0, limit = ssLog(last - first);;) {
        if ((last - first) <= INSERTIONSORT_THRESHOLD) {
            if (1 < (last - first)) ssInsertionSort(D, PA, first, last, depth);
            if (ssize == 0) return;
            ssize--; first = stack[ssize].a; last = stack[ssize].b; depth = stack[ssize].c; limit = stack[ssize].d;
            continue;
        }
        Td = depth;
        if (limit-- == 0) ssHeapSort(D, Td, PA, first, last - first);
        if (limit < 0) {
            for (a = first + 1, v = SSV(first); a < last; ++a) {
                if ((x = SSV(a)) != v) {
                    if (1 < (a - first)) break;
                    v = x;
                    first = a;
                }
            }
            if (T(Td + SA[PA + SA[first]] - 1) < v) first = ssSubstringPartition(D, PA, first, a, depth);
            if ((a - first) <= (last - a)) {
                if (1 < (a - first)) {
                    PUSH(a, last, depth, -1);
                    last = a; depth += 1; limit = ssLog(a - first);
                } else {
                    first = a; limit = -1;
                }
            } else {
                if (1 < (last - a)) {
                    PUSH(first, a, depth + 1, ssLog(a - first));
                    first = a; limit = -1;
                } else {
                    last = a; depth += 1; limit = ssLog(a - first);
                }
            }
            continue;
        }
        a = ssPivot(D, Td, PA, first, last);
        v = SSV(a);
        SWAP(first, a);
        for (b = first; (++b < last) && ((x = SSV(b)) == v);) {}
        if (((a = b) < last) && (x < v)) {
            for (; (++b < last) && ((x = SSV(b)) <= v);) {
                if (x == v) { SWAP(b, a); ++a; }
            }
        }
        for (c = last; (b < --c) && ((x = SSV(c)) == v);) {}
        if ((b < (d = c)) && (x > v)) {
            for (; (b < --c) && ((x = SSV(c)) >= v);) {
                if (x == v) { SWAP(c, d); --d; }
            }
        }
        for (; b < c;) {
            SWAP(b, c);
            for (; (++b < c) && ((x = SSV(b)) <= v);) {
                if (x == v) { SWAP(b, a); ++a; }
            }
            for (; (b < --c) && ((x = SSV(c)) >= v);) {
                if (x == v) { SWAP(c, d); --d; }
            }
        }
        if (a <= d) {
            c = b - 1;
            if ((s = a - first) > (t = b - a)) s = t;
            for (e = first, f = b - s; 0 < s; --s, ++e, ++f) SWAP(e, f);
            if ((s = d - c) > (t = last - d - 1)) s = t;
            for (e = b, f = last - s; 0 < s; --s, ++e, ++f) SWAP(e, f);
            a = first + (b - a);
            c = last - (d - c);
            b = (v <= T(Td + SA[PA + SA[a]] - 1)) ? a : ssSubstringPartition(D, PA, a, c, depth);
            if ((a - first) <= (last - c)) {
                if ((last - c) <= (c - b)) {
                    PUSH(b, c, depth + 1, ssLog(c - b));
                    PUSH(c, last, depth, limit);
                    last = a;
                } else if ((a - first) <= (c - b)) {
                    PUSH(c, last, depth, limit);
                    PUSH(b, c, depth + 1, ssLog(c - b));
                    last = a;
                } else {
                    PUSH(c, last, depth, limit);
                    PUSH(first, a, depth, limit);
                    first = b; last = c; depth += 1; limit = ssLog(c - b);
                }
            } else {
                if ((a - first) <= (c - b)) {
                    PUSH(b, c, depth + 1, ssLog(c - b));
                    PUSH(first, a, depth, limit);
                    first = c;
                } else if ((last - c) <= (c - b)) {
                    PUSH(first, a, depth, limit);
                    PUSH(b, c, depth + 1, ssLog(c - b));
                    first = c;
                } else {
                    PUSH(first, a, depth, limit);
                    PUSH(c, last, depth, limit);
                    first = b; last = c; depth += 1; limit = ssLog(c - b);
                }
            }
        } else {
            limit += 1;
            if (T(Td + SA[PA + SA[first]] - 1) < v) {
                first = ssSubstringPartition(D, PA, first, last, depth);
                limit = ssLog(last - first);
            }
            depth += 1;
        }
    }
}

/* buffers: `buf` is either SA itself or the 256-entry temp buffer (D:1387-1394, 3220-3227) */
#define BUF(i) bufp[(i)]

/* D:892-912 */
static __device__ void ssBlockSwap(int *a1, int first1, int *a2, int first2, int size) {
    int a, b, i;
    for (i = size, a = first1, b = first2; 0 < i; --i, ++a, ++b) { int t = a1[a]; a1[a] = a2[b]; a2[b] = t; }
}

/* D:923-1001 */
static __device__ void ssMergeForward(dss_t *D, int PA, int *bufp, int bufoffset, int first, int middle, int last, int depth) {
    int bufend, i, j, k, t, r;
    bufend = bufoffset + (middle - first) - 1;
    ssBlockSwap(bufp, bufoffset, SA, first, middle - first);
    for (t = SA[first], i = first, j = bufoffset, k = middle;;) {
        r = ssCompare(D, PA + BUF(j), PA + SA[k], depth);
        if (r < 0) {
            do {
                SA[i++] = BUF(j);
                if (bufend <= j) { BUF(j) = t; return; }
                BUF(j++) = SA[i];
            } while (BUF(j) < 0);
        } else if (r > 0) {
            do {
                SA[i++] = SA[k]; SA[k++] = SA[i];
                if (last <= k) {
                    while (j < bufend) { SA[i++] = BUF(j); BUF(j++) = SA[i]; }
                    SA[i] = BUF(j); BUF(j) = t;
                    return;
                }
            } while (SA[k] < 0);
        } else {
            SA[k] = ~SA[k];
            do {
                SA[i++] = BUF(j);
                if (bufend <= j) { BUF(j) = t; return; }
                BUF(j++) = SA[i];
            } while (BUF(j) < 0);
            do {
                SA[i++] = SA[k]; SA[k++] = SA[i];
                if (last <= k) {
                    while (j < bufend) { SA[i++] = BUF(j); BUF(j++) = SA[i]; }
                    SA[i] = BUF(j); BUF(j) = t;
                    return;
                }
            } while (SA[k] < 0);
        }
    }
}

/* D:1014-1165 */
static __device__ void ssMergeBackward(dss_t *D, int PA, int *bufp, int bufoffset, int first, int middle, int last, int depth) {
    int p1, p2, bufend, i, j, k, t, r, x;
    bufend = bufoffset + (last - middle);
    ssBlockSwap(bufp, bufoffset, SA, middle, last - middle);
    x = 0;
    if (BUF(bufend - 1) < 0) { x |= 1; p1 = PA + ~BUF(bufend - 1); } else { p1 = PA + BUF(bufend - 1); }
    if (SA[middle - 1] < 0) { x |= 2; p2 = PA + ~SA[middle - 1]; } else { p2 = PA + SA[middle - 1]; }
    for (t = SA[last - 1], i = last - 1, j = bufend - 1, k = middle - 1;;) {
        r = ssCompare(D, p1, p2, depth);
        if (r > 0) {
            if ((x & 1) != 0) {
                do { SA[i--] = BUF(j); BUF(j--) = SA[i]; } while (BUF(j) < 0);
                x ^= 1;
            }
            SA[i--] = BUF(j);
            if (j <= bufoffset) { BUF(j) = t; return; }
            BUF(j--) = SA[i];
            if (BUF(j) < 0) { x |= 1; p1 = PA + ~BUF(j); } else { p1 = PA + BUF(j); }
        } else if (r < 0) {
            if ((x & 2) != 0) {
                do { SA[i--] = SA[k]; SA[k--] = SA[i]; } while (SA[k] < 0);
                x ^= 2;
            }
            SA[i--] = SA[k]; SA[k--] = SA[i];
            if (k < first) {
                while (bufoffset < j) { SA[i--] = BUF(j); BUF(j--) = SA[i]; }
                SA[i] = BUF(j); BUF(j) = t;
                return;
            }
            if (SA[k] < 0) { x |= 2; p2 = PA + ~SA[k]; } else { p2 = PA + SA[k]; }
        } else {
            if ((x & 1) != 0) {
                do { SA[i--] = BUF(j); BUF(j--) = SA[i]; } while (BUF(j) < 0);
                x ^= 1;
            }
            SA[i--] = ~BUF(j);
            if (j <= bufoffset) { BUF(j) = t; return; }
            BUF(j--) = SA[i];
            if ((x & 2) != 0) {
                do { SA[i--] = SA[k]; SA[k--] = SA[i]; } while (SA[k] < 0);
                x ^= 2;
            }
            SA[i--] = SA[k]; SA[k--] = SA[i];
            if (k < first) {
                while (bufoffset < j) { SA[i--] = BUF(j); BUF(j--) = SA[i]; }
                SA[i] = BUF(j); BUF(j) = t;
                return;
            }
            if (BUF(j) < 0) { x |= 1; p1 = PA + ~BUF(j); } else { p1 = PA + BUF(j); }
            if (SA[k] < 0) { x |= 2; p2 = PA + ~SA[k]; } else { p2 = PA + SA[k]; }
        }
    }
}

/* D:1171-1173 */
static __device__ int getIDX(int a) { return (0 <= a) ? a : ~a; }

/* D:1180-1188 */
static __device__ void ssMergeCheckEqual(dss_t *D, int PA, int depth, int a) {
    if ((0 <= SA[a]) && (ssCompare(D, PA + getIDX(SA[a - 1]), PA + SA[a], depth) == 0)) SA[a] = ~SA[a];
}

/* D:1200-1353 */
static __device__ void ssMerge(dss_t *D, int PA, int first, int middle, int last, int *bufp, int bufoffset, int bufsize, int depth) {
    stack_entry_t stack[STACK_SIZE];
    int i, j, m, len, half, ssize, check, next;
    for (check = 0, ssize = 0;;) {
        if ((last - middle) <= bufsize) {
            if ((first < middle) && (middle < last)) ssMergeBackward(D, PA, bufp, bufoffset, first, middle, last, depth);
            if ((check & 1) != 0) ssMergeCheckEqual(D, PA, depth, first);
            if ((check & 2) != 0) ssMergeCheckEqual(D, PA, depth, last);
            if (ssize == 0) return;
            ssize--; first = stack[ssize].a; middle = stack[ssize].b; last = stack[ssize].c; check = stack[ssize].d;
            continue;
        }
        if ((middle - first) <= bufsize) {
            if (first < middle) ssMergeForward(D, PA, bufp, bufoffset, first, middle, last, depth);
            if ((check & 1) != 0) ssMergeCheckEqual(D, PA, depth, first);
            if ((check & 2) != 0) ssMergeCheckEqual(D, PA, depth, last);
            if (ssize == 0) return;
            ssize--; first = stack[ssize].a; middle = stack[ssize].b; last = stack[ssize].c; check = stack[ssize].d;
            continue;
        }
        for (m = 0, len = ((middle - first) < (last - middle) ? (middle - first) : (last - middle)), half = len >> 1;
             0 < len; len = half, half >>= 1) {
            if (ssCompare(D, PA + getIDX(SA[middle + m + half]), PA + getIDX(SA[middle - m - half - 1]), depth) < 0) {
                m += half + 1;
                half -= (len & 1) ^ 1;
            }
        }
        if (0 < m) {
            ssBlockSwap(SA, middle - m, SA, middle, m);
            i = j = middle;
            next = 0;
            if ((middle + m) < last) {
                if (SA[middle + m] < 0) {
                    for (; SA[i - 1] < 0; --i) {}
                    SA[middle + m] = ~SA[middle + m];
                }
                for (j = middle; SA[j] < 0; ++j) {}
                next = 1;
            }
            if ((i - first) <= (last - j)) {
                PUSH(j, middle + m, last, (check & 2) | (next & 1));
                middle -= m; last = i; check = (check & 1);
            } else {
                if ((i == middle) && (middle == j)) next <<= 1;
                PUSH(first, middle - m, i, (check & 1) | (next & 2));
                first = j; middle += m; check = (check & 2) | (next & 1);
            }
        } else {
            if ((check & 1) != 0) ssMergeCheckEqual(D, PA, depth, first);
            ssMergeCheckEqual(D, PA, depth, middle);
            if ((check & 2) != 0) ssMergeCheckEqual(D, PA, depth, last);
            if (ssize == 0) return;
            ssize--; first = stack[ssize].a; middle = stack[ssize].b; last = stack[ssize].c; check = stack[ssize].d;
        }
    }
}

/* D:1367-1467 */
static __device__ void subStringSort(dss_t *D, int PA, int first, int last, int *bufp, int bufoffset, int bufsize,
                          int depth, int lastsuffix, int size) {
    int a, b, *curbuf, curbufoffset, i, j, k, curbufsize;
    if (lastsuffix) ++first;
    for (a = first, i = 0; (a + SS_BLOCKSIZE) < last; a += SS_BLOCKSIZE, ++i) {
        ssMultiKeyIntroSort(D, PA, a, a + SS_BLOCKSIZE, depth);
        curbuf = SA;
        curbufoffset = a + SS_BLOCKSIZE;
        curbufsize = last - (a + SS_BLOCKSIZE);
        if (curbufsize <= bufsize) { curbufsize = bufsize; curbuf = bufp; curbufoffset = bufoffset; }
        for (b = a, k = SS_BLOCKSIZE, j = i; (j & 1) != 0; b -= k, k <<= 1, j = (int)((unsigned)j >> 1))
            ssMerge(D, PA, b - k, b, b + k, curbuf, curbufoffset, curbufsize, depth);
    }
    ssMultiKeyIntroSort(D, PA, a, last, depth);
    for (k = SS_BLOCKSIZE; i != 0; k <<= 1, i >>= 1) {
        if ((i & 1) != 0) {
            ssMerge(D, PA, a - k, a, last, bufp, bufoffset, bufsize, depth);
            a -= k;
        }
    }
    if (lastsuffix) {
        int r;
        for (a = first, i = SA[first - 1], r = 1;
             (a < last) && ((SA[a] < 0) || (0 < (r = ssCompareLast(D, PA, PA + i, PA + SA[a], depth, size))));
             ++a)
            SA[a - 1] = SA[a];
        if (r == 0) SA[a] = ~SA[a];
        SA[a - 1] = i;
    }
}

/* ------------------------------------------------------------------ rank sort (tr*, ls*) */
/* D:1476-1480 (bzip2 patch: cyclic wrap) */
static __device__ int trGetC(dss_t *D, int ISA, int ISAd, int ISAn, int p) {
    return ((ISAd + p) < ISAn) ? SA[ISAd + p] : SA[ISA + ((ISAd - ISA + p) % (ISAn - ISA))];
}
#define TRC(p) trGetC(D, ISA, ISAd, ISAn, (p))

/* D:1491-1530 (Java header at :1500; v, c fixed before the loop) */
static __device__ void trFixdown(dss_t *D, int ISA, int ISAd, int ISAn, int sa, int i, int size) {
    int j, k, v, c, d, e;
    for (v = SA[sa + i], c = TRC(v); (j = 2 * i + 1) < size; SA[sa + i] = SA[sa + k], i = k) {
        k = j++;
        d = TRC(SA[sa + k]);
        if (d < (e = TRC(SA[sa + j]))) { k = j; d = e; }
        if (d <= c) break;
    }
    SA[sa + i] = v;
}

/* D:1540-1571 */
static __device__ void trHeapSort(dss_t *D, int ISA, int ISAd, int ISAn, int sa, int size) {
    int i, m, t;
    m = size;
    if ((size % 2) == 0) {
        m--;
        if (TRC(SA[sa + (m / 2)]) < TRC(SA[sa + m])) SWAP(sa + m, sa + (m / 2));
    }
    for (i = m / 2 - 1; 0 <= i; --i) trFixdown(D, ISA, ISAd, ISAn, sa, i, m);
    if ((size % 2) == 0) { SWAP(sa + 0, sa + m); trFixdown(D, ISA, ISAd, ISAn, sa, 0, m); }
    for (i = m - 1; 0 < i; --i) {
        t = SA[sa + 0]; SA[sa + 0] = SA[sa + i];
        trFixdown(D, ISA, ISAd, ISAn, sa, 0, i);
        SA[sa + i] = t;
    }
}

/* D:1581-1621 */
static __device__ void trInsertionSort(dss_t *D, int ISA, int ISAd, int ISAn, int first, int last) {
    int a, b, t, r;
    for (a = first + 1; a < last; ++a) {
        for (t = SA[a], b = a - 1; 0 > (r = TRC(t) - TRC(SA[b]));) {
            do { SA[b + 1] = SA[b]; } while ((first <= --b) && (SA[b] < 0));
            if (b < first) break;
        }
        if (r == 0) SA[b] = ~SA[b];
        SA[b + 1] = t;
    }
}

/* D:1628-1632 */
static __device__ int trLog(int n) {
    return ((n & 0xffff0000) != 0)
               ? (((n & 0xff000000) != 0) ? 24 + log2table[(n >> 24) & 0xff] : 16 + log2table[(n >> 16) & 0xff])
               : (((n & 0x0000ff00) != 0) ? 8 + log2table[(n >> 8) & 0xff] : 0 + log2table[(n >> 0) & 0xff]);
}

/* D:1644-1669 */
static __device__ int trMedian3(dss_t *D, int ISA, int ISAd, int ISAn, int v1, int v2, int v3) {
    int a1 = TRC(SA[v1]), a2 = TRC(SA[v2]), a3 = TRC(SA[v3]);
    if (a1 > a2) { int t = v1; v1 = v2; v2 = t; int tt = a1; a1 = a2; a2 = tt; }
    if (a2 > a3) { if (a1 > a3) return v1; return v3; }
    return v2;
}
/* D:1683-1755 */
static __device__ int trMedian5(dss_t *D, int ISA, int ISAd, int ISAn, int v1, int v2, int v3, int v4, int v5) {
    int a1 = TRC(SA[v1]), a2 = TRC(SA[v2]), a3 = TRC(SA[v3]), a4 = TRC(SA[v4]), a5 = TRC(SA[v5]), t, tt;
    if (a2 > a3) { t = v2; v2 = v3; v3 = t; tt = a2; a2 = a3; a3 = tt; }
    if (a4 > a5) { t = v4; v4 = v5; v5 = t; tt = a4; a4 = a5; a5 = tt; }
    if (a2 > a4) { t = v2; v2 = v4; v4 = t; tt = a2; a2 = a4; a4 = tt;
                   t = v3; v3 = v5; v5 = t; tt = a3; a3 = a5; a5 = tt; }
    if (a1 > a3) { t = v1; v1 = v3; v3 = t; tt = a1; a1 = a3; a3 = tt; }
    if (a1 > a4) { t = v1; v1 = v4; v4 = t; tt = a1; a1 = a4; a4 = tt;
                   t = v3; v3 = v5; v5 = t; tt = a3; a3 = a5; a5 = tt; }
    if (a3 > a4) return v4;
    return v3;
}
/* D:1766-1794 */
static __device__ int trPivot(dss_t *D, int ISA, int ISAd, int ISAn, int first, int last) {
    int middle, t;
    t = last - first;
    middle = first + t / 2;
    if (t <= 512) {
        if (t <= 32) return trMedian3(D, ISA, ISAd, ISAn, first, middle, last - 1);
        t >>= 2;
        return trMedian5(D, ISA, ISAd, ISAn, first, first + t, middle, last - 1 - t, last - 1);
    }
    t >>= 3;
    return trMedian3(D, ISA, ISAd, ISAn,
                     trMedian3(D, ISA, ISAd, ISAn, first, first + t, first + (t << 1)),
                     trMedian3(D, ISA, ISAd, ISAn, middle - t, middle, middle + t),
                     trMedian3(D, ISA, ISAd, ISAn, last - 1 - (t << 1), last - 1 - t, last - 1));
}

/* D:1802-1840 (Java loop at :1807-1814: the inner ++a steer the iteration) */
static __device__ void lsUpdateGroup(dss_t *D, int ISA, int first, int last) {
    int a, b, t;
    for (a = first; a < last; ++a) {
        if (0 <= SA[a]) {
            b = a;
            do { SA[ISA + SA[a]] = a; } while ((++a < last) && (0 <= SA[a]));
            SA[b] = b - a;
            if (last <= a) break;
        }
        b = a;
        do { SA[a] = ~SA[a]; } while (SA[++a] < 0);
        t = a;
        do { SA[ISA + SA[b]] = t; } while (++b <= a);
    }
}

/* D:1851-2160 */
static __device__ void lsIntroSort(dss_t *D, int ISA, int ISAd, int ISAn, int first, int last) {
    stack_entry_t stack[STACK_SIZE];
    int a, b, c, d, e, f, s, t, limit, v, x = 0, ssize;
    for (ssize = 0, limit = trLog(last - first);;) {
        if ((last - first) <= INSERTIONSORT_THRESHOLD) {
            if (1 < (last - first)) {
                trInsertionSort(D, ISA, ISAd, ISAn, first, last);
                lsUpdateGroup(D, ISA, first, last);
            } else if ((last - first) == 1) {
                SA[first] = -1;
            }
            if (ssize == 0) return;
            ssize--; first = stack[ssize].a; last = stack[ssize].b; limit = stack[ssize].c;
            continue;
        }
        if (limit-- == 0) {
            trHeapSort(D, ISA, ISAd, ISAn, first, last - first);
            for (a = last - 1; first < a; a = b) {
                for (x = TRC(SA[a]), b = a - 1; (first <= b) && (TRC(SA[b]) == x); --b) SA[b] = ~SA[b];
            }
            lsUpdateGroup(D, ISA, first, last);
            if (ssize == 0) return;
            ssize--; first = stack[ssize].a; last = stack[ssize].b; limit = stack[ssize].c;
            continue;
        }
        a = trPivot(D, ISA, ISAd, ISAn, first, last);
        SWAP(first, a);
        v = TRC(SA[first]);
        for (b = first; (++b < last) && ((x = TRC(SA[b])) == v);) {}
        if (((a = b) < last) && (x < v)) {
            for (; (++b < last) && ((x = TRC(SA[b])) <= v);) {
                if (x == v) { SWAP(b, a); ++a; }
            }
        }
        for (c = last; (b < --c) && ((x = TRC(SA[c])) == v);) {}
        if ((b < (d = c)) && (x > v)) {
            for (; (b < --c) && ((x = TRC(SA[c])) >= v);) {
                if (x == v) { SWAP(c, d); --d; }
            }
        }
        for (; b < c;) {
            SWAP(b, c);
            for (; (++b < c) && ((x = TRC(SA[b])) <= v);) {
                if (x == v) { SWAP(b, a); ++a; }
            }
            for (; (b < --c) && ((x = TRC(SA[c])) >= v);) {
                if (x == v) { SWAP(c, d); --d; }
            }
        }
        if (a <= d) {
            c = b - 1;
            if ((s = a - first) > (t = b - a)) s = t;
            for (e = first, f = b - s; 0 < s; --s, ++e, ++f) SWAP(e, f);
            if ((s = d - c) > (t = last - d - 1)) s = t;
            for (e = b, f = last - s; 0 < s; --s, ++e, ++f) SWAP(e, f);
            a = first + (b - a);
            b = last - (d - c);
            for (c = first, v = a - 1; c < a; ++c) SA[ISA + SA[c]] = v;
            if (b < last) {
                for (c = a, v = b - 1; c < b; ++c) SA[ISA + SA[c]] = v;
            }
            if ((b - a) == 1) SA[a] = -1;
            if ((a - first) <= (last - b)) {
                if (first < a) { PUSH(b, last, limit, 0); last = a; }
                else first = b;
            } else {
                if (b < last) { PUSH(first, a, limit, 0); first = b; }
                else last = a;
            }
        } else {
            if (ssize == 0) return;
            ssize--; first = stack[ssize].a; last = stack[ssize].b; limit = stack[ssize].c;
        }
    }
}

/* D:2168-2222 */
static __device__ void lsSort(dss_t *D, int ISA, int n, int depth) {
    int ISAd, first, last, i, t, skip;
    for (ISAd = ISA + depth; -n < SA[0]; ISAd += (ISAd - ISA)) {
        first = 0;
        skip = 0;
        do {
            if ((t = SA[first]) < 0) {
                first -= t;
                skip += t;
            } else {
                if (skip != 0) { SA[first + skip] = skip; skip = 0; }
                last = SA[ISA + t] + 1;
                lsIntroSort(D, ISA, ISAd, ISA + n, first, last);
                first = last;
            }
        } while (first < n);
        if (skip != 0) SA[first + skip] = skip;
        if (n < (ISAd - ISA)) {
            first = 0;
            do {
                if ((t = SA[first]) < 0) {
                    first -= t;
                } else {
                    last = SA[ISA + t] + 1;
                    for (i = first; i < last; ++i) SA[ISA + SA[i]] = i;
                    first = last;
                }
            } while (first < n);
            break;
        }
    }
}

/* D:2233-2421 (BZip2PartitionResult.swift:29-48); Java loop headers */
static __device__ void trPartition(dss_t *D, int ISA, int ISAd, int ISAn, int first, int last, int v, int *pa, int *pb) {
    int a, b, c, d, e, f, t, s, x = 0;
    for (b = first - 1; (++b < last) && ((x = TRC(SA[b])) == v);) {}
    if (((a = b) < last) && (x < v)) {
        for (; (++b < last) && ((x = TRC(SA[b])) <= v);) {
            if (x == v) { SWAP(b, a); ++a; }
        }
    }
    for (c = last; (b < --c) && ((x = TRC(SA[c])) == v);) {}
    if ((b < (d = c)) && (x > v)) {
        for (; (b < --c) && ((x = TRC(SA[c])) >= v);) {
            if (x == v) { SWAP(c, d); --d; }
        }
    }
    for (; b < c;) {
        SWAP(b, c);
        for (; (++b < c) && ((x = TRC(SA[b])) <= v);) {
            if (x == v) { SWAP(b, a); ++a; }
        }
        for (; (b < --c) && ((x = TRC(SA[c])) >= v);) {
            if (x == v) { SWAP(c, d); --d; }
        }
    }
    if (a <= d) {
        c = b - 1;
        if ((s = a - first) > (t = b - a)) s = t;
        for (e = first, f = b - s; 0 < s; --s, ++e, ++f) SWAP(e, f);
        if ((s = d - c) > (t = last - d - 1)) s = t;
        for (e = b, f = last - s; 0 < s; --s, ++e, ++f) SWAP(e, f);
        first += (b - a);
        last -= (d - c);
    }
    *pa = first;
    *pb = last;
}

/* D:2433-2477 (Java headers at :2443, :2458; d grows inside the first loop) */
static __device__ void trCopy(dss_t *D, int ISA, int ISAn, int first, int a, int b, int last, int depth) {
    int c, d, e, s, v;
    v = b - 1;
    for (c = first, d = a - 1; c <= d; ++c) {
        if ((s = SA[c] - depth) < 0) s += ISAn - ISA;
        if (SA[ISA + s] == v) { SA[++d] = s; SA[ISA + s] = d; }
    }
    for (c = last - 1, e = d + 1, d = b; e < d; --c) {
        if ((s = SA[c] - depth) < 0) s += ISAn - ISA;
        if (SA[ISA + s] == v) { SA[--d] = s; SA[ISA + s] = d; }
    }
}

static __device__ int saReadPastEnd(dss_t *D, int idx) {
    if (idx >= D->sa_len) { D->trap = 1; return 0; }
    return SA[idx];
}

/* D:2489-3008 */
static __device__ void trIntroSort(dss_t *D, int ISA, int ISAd, int ISAn, int first, int last, tr_budget_t *budget, int size) {
    stack_entry_t stack[STACK_SIZE];
    int a, b, c, d, e, f, s, t, v, x = 0, limit, next, ssize;
    for (ssize = 0, limit = trLog(last - first);;) {
        if (limit < 0) {
            if (limit == -1) {
                if (!budget_update(budget, size, last - first)) break;
                trPartition(D, ISA, ISAd - 1, ISAn, first, last, last - 1, &a, &b);
                if ((first < a) || (b < last)) {
                    if (a < last) {
                        for (c = first, v = a - 1; c < a; ++c) SA[ISA + SA[c]] = v;
                    }
                    if (b < last) {
                        for (c = a, v = b - 1; c < b; ++c) SA[ISA + SA[c]] = v;
                    }
                    PUSH(0, a, b, 0);
                    PUSH(ISAd - 1, first, last, -2);
                    if ((a - first) <= (last - b)) {
                        if (1 < (a - first)) {
                            PUSH(ISAd, b, last, trLog(last - b));
                            last = a; limit = trLog(a - first);
                        } else if (1 < (last - b)) {
                            first = b; limit = trLog(last - b);
                        } else {
                            if (ssize == 0) return;
                            ssize--; ISAd = stack[ssize].a; first = stack[ssize].b; last = stack[ssize].c; limit = stack[ssize].d;
                        }
                    } else {
                        if (1 < (last - b)) {
                            PUSH(ISAd, first, a, trLog(a - first));
                            first = b; limit = trLog(last - b);
                        } else if (1 < (a - first)) {
                            last = a; limit = trLog(a - first);
                        } else {
                            if (ssize == 0) return;
                            ssize--; ISAd = stack[ssize].a; first = stack[ssize].b; last = stack[ssize].c; limit = stack[ssize].d;
                        }
                    }
                } else {
                    for (c = first; c < last; ++c) SA[ISA + SA[c]] = c;
                    if (ssize == 0) return;
                    ssize--; ISAd = stack[ssize].a; first = stack[ssize].b; last = stack[ssize].c; limit = stack[ssize].d;
                }
            } else if (limit == -2) {
                ssize--;
                a = stack[ssize].b;
                b = stack[ssize].c;
                trCopy(D, ISA, ISAn, first, a, b, last, ISAd - ISA);
                if (ssize == 0) return;
                ssize--; ISAd = stack[ssize].a; first = stack[ssize].b; last = stack[ssize].c; limit = stack[ssize].d;
            } else {
                if (0 <= SA[first]) {
                    a = first;
                    do { SA[ISA + SA[a]] = a; } while ((++a < last) && (0 <= SA[a]));
                    first = a;
                }
                if (first < last) {
                    a = first;
                    do { SA[a] = ~SA[a]; } while (SA[++a] < 0);
                    // direct index, not trGetC (D:2589-2592): on cyclic input ISAd + SA[a] can pass ISAn and
                    // even the end of the reference's array (blockSize + 1 ints, zero-initialised,
                    // S/BZip2BlockCompressor.swift:318), where the reference fails with an index
                    // error; here that case reads 0 and is counted in ctl[14].
                    next = (SA[ISA + SA[a]] != saReadPastEnd(D, ISAd + SA[a])) ? trLog(a - first + 1) : -1;
                    if (++a < last) {
                        for (b = first, v = a - 1; b < a; ++b) SA[ISA + SA[b]] = v;
                    }
                    if ((a - first) <= (last - a)) {
                        PUSH(ISAd, a, last, -3);
                        ISAd += 1; last = a; limit = next;
                    } else {
                        if (1 < (last - a)) {
                            PUSH(ISAd + 1, first, a, next);
                            first = a; limit = -3;
                        } else {
                            ISAd += 1; last = a; limit = next;
                        }
                    }
                } else {
                    if (ssize == 0) return;
                    ssize--; ISAd = stack[ssize].a; first = stack[ssize].b; last = stack[ssize].c; limit = stack[ssize].d;
                }
            }
            continue;
        }
        if ((last - first) <= INSERTIONSORT_THRESHOLD) {
            if (!budget_update(budget, size, last - first)) break;
            trInsertionSort(D, ISA, ISAd, ISAn, first, last);
            limit = -3;
            continue;
        }
        if (limit-- == 0) {
            if (!budget_update(budget, size, last - first)) break;
            trHeapSort(D, ISA, ISAd, ISAn, first, last - first);
            for (a = last - 1; first < a; a = b) {
                for (x = TRC(SA[a]), b = a - 1; (first <= b) && (TRC(SA[b]) == x); --b) SA[b] = ~SA[b];
            }
            limit = -3;
            continue;
        }
        a = trPivot(D, ISA, ISAd, ISAn, first, last);
        SWAP(first, a);
        v = TRC(SA[first]);
        for (b = first; (++b < last) && ((x = TRC(SA[b])) == v);) {}
        if (((a = b) < last) && (x < v)) {
            for (; (++b < last) && ((x = TRC(SA[b])) <= v);) {
                if (x == v) { SWAP(b, a); ++a; }
            }
        }
        for (c = last; (b < --c) && ((x = TRC(SA[c])) == v);) {}
        if ((b < (d = c)) && (x > v)) {
            for (; (b < --c) && ((x = TRC(SA[c])) >= v);) {
                if (x == v) { SWAP(c, d); --d; }
            }
        }
        for (; b < c;) {
            SWAP(b, c);
            for (; (++b < c) && ((x = TRC(SA[b])) <= v);) {
                if (x == v) { SWAP(b, a); ++a; }
            }
            for (; (b < --c) && ((x = TRC(SA[c])) >= v);) {
                if (x == v) { SWAP(c, d); --d; }
            }
        }
        if (a <= d) {
            c = b - 1;
            if ((s = a - first) > (t = b - a)) s = t;
            for (e = first, f = b - s; 0 < s; --s, ++e, ++f) SWAP(e, f);
            if ((s = d - c) > (t = last - d - 1)) s = t;
            for (e = b, f = last - s; 0 < s; --s, ++e, ++f) SWAP(e, f);
            a = first + (b - a);
            b = last - (d - c);
            next = (SA[ISA + SA[a]] != v) ? trLog(b - a) : -1;
            for (c = first, v = a - 1; c < a; ++c) SA[ISA + SA[c]] = v;
            if (b < last) {
                for (c = a, v = b - 1; c < b; ++c) SA[ISA + SA[c]] = v;
            }
            if ((a - first) <= (last - b)) {
                if ((last - b) <= (b - a)) {
                    if (1 < (a - first)) {
                        PUSH(ISAd + 1, a, b, next);
                        PUSH(ISAd, b, last, limit);
                        last = a;
                    } else if (1 < (last - b)) {
                        PUSH(ISAd + 1, a, b, next);
                        first = b;
                    } else if (1 < (b - a)) {
                        ISAd += 1; first = a; last = b; limit = next;
                    } else {
                        if (ssize == 0) return;
                        ssize--; ISAd = stack[ssize].a; first = stack[ssize].b; last = stack[ssize].c; limit = stack[ssize].d;
                    }
                } else if ((a - first) <= (b - a)) {
                    if (1 < (a - first)) {
                        PUSH(ISAd, b, last, limit);
                        PUSH(ISAd + 1, a, b, next);
                        last = a;
                    } else if (1 < (b - a)) {
                        PUSH(ISAd, b, last, limit);
                        ISAd += 1; first = a; last = b; limit = next;
                    } else {
                        first = b;
                    }
                } else {
                    if (1 < (b - a)) {
                        PUSH(ISAd, b, last, limit);
                        PUSH(ISAd, first, a, limit);
                        ISAd += 1; first = a; last = b; limit = next;
                    } else {
                        PUSH(ISAd, b, last, limit);
                        last = a;
                    }
                }
            } else {
                if ((a - first) <= (b - a)) {
                    if (1 < (last - b)) {
                        PUSH(ISAd + 1, a, b, next);
                        PUSH(ISAd, first, a, limit);
                        first = b;
                    } else if (1 < (a - first)) {
                        PUSH(ISAd + 1, a, b, next);
                        last = a;
                    } else if (1 < (b - a)) {
                        ISAd += 1; first = a; last = b; limit = next;
                    } else {
                        PUSH(ISAd, first, last, limit);
                    }
                } else if ((last - b) <= (b - a)) {
                    if (1 < (last - b)) {
                        PUSH(ISAd, first, a, limit);
                        PUSH(ISAd + 1, a, b, next);
                        first = b;
                    } else if (1 < (b - a)) {
                        PUSH(ISAd, first, a, limit);
                        ISAd += 1; first = a; last = b; limit = next;
                    } else {
                        last = a;
                    }
                } else {
                    if (1 < (b - a)) {
                        PUSH(ISAd, first, a, limit);
                        PUSH(ISAd, b, last, limit);
                        ISAd += 1; first = a; last = b; limit = next;
                    } else {
                        PUSH(ISAd, first, a, limit);
                        first = b;
                    }
                }
            }
        } else {
            if (!budget_update(budget, size, last - first)) break; /* reference BUGFIX :2996 */
            limit += 1; ISAd += 1;
        }
    }
    for (s = 0; s < ssize; ++s) {
        if (stack[s].d == -3) lsUpdateGroup(D, ISA, stack[s].b, stack[s].c);
    }
}

/* D:3015-3047 */
static __device__ void trSort(dss_t *D, int ISA, int n, int depth) {
    int first = 0, last, t;
    if (-n < SA[0]) {
        tr_budget_t budget = {n, trLog(n) * 2 / 3 + 1};
        do {
            if ((t = SA[first]) < 0) {
                first -= t;
            } else {
                last = SA[ISA + t] + 1;
                if (1 < (last - first)) {
                    trIntroSort(D, ISA, ISA + depth, ISA + n, first, last, &budget, n);
                    if (budget.chance == 0) {
                        /* Switch to Larsson-Sadakane sorting algorithm. */
                        if (0 < first) SA[0] = -first;
                        lsSort(D, ISA, n, depth);
                        break;
                    }
                }
                first = last;
            }
        } while (first < n);
    }
}

/* D:3055-3068 */
#define BUCKET_B(c0, c1) (((c1) << 8) | (c0))
#define BUCKET_BSTAR(c0, c1) (((c0) << 8) | (c1))

/* D:3076-3362 */
static __device__ int sortTypeBstar(dss_t *D, int *bucketA, int *bucketB) {
    const int n = D->n;
    int tempbuf[256];
    int *bufp, PAb, ISAb, bufoffset, i, j, k, t, m, bufsize, c0, c1, flag, ti, t0, ti1;
    for (i = 0; i < 256; i++) tempbuf[i] = 0;
    for (i = 1, flag = 1; i < n; ++i) {
        if (T(i - 1) != T(i)) {
            if (T(i - 1) > T(i)) flag = 0;
            break;
        }
    }
    i = n - 1;
    m = n;
    if (((ti = T(i)) < (t0 = T(0))) || ((T(i) == T(0)) && (flag != 0))) {
        if (flag == 0) {
            ++bucketB[BUCKET_BSTAR(ti, t0)];
            SA[--m] = i;
        } else {
            ++bucketB[BUCKET_B(ti, t0)];
        }
        for (--i; (0 <= i) && ((ti = T(i)) <= (ti1 = T(i + 1))); --i) ++bucketB[BUCKET_B(ti, ti1)];
    }
    for (; 0 <= i;) {
        do { ++bucketA[T(i)]; } while ((0 <= --i) && (T(i) >= T(i + 1)));
        if (0 <= i) {
            ++bucketB[BUCKET_BSTAR(T(i), T(i + 1))];
            SA[--m] = i;
            for (--i; (0 <= i) && ((ti = T(i)) <= (ti1 = T(i + 1))); --i) ++bucketB[BUCKET_B(ti, ti1)];
        }
    }
    m = n - m;
    if (m == 0) {
        for (i = 0; i < n; ++i) SA[i] = i;
        return 0;
    }
    for (c0 = 0, i = -1, j = 0; c0 < 256; ++c0) {
        t = i + bucketA[c0];
        bucketA[c0] = i + j;
        i = t + bucketB[BUCKET_B(c0, c0)];
        for (c1 = c0 + 1; c1 < 256; ++c1) {
            j += bucketB[BUCKET_BSTAR(c0, c1)];
            bucketB[(c0 << 8) | c1] = j;
            i += bucketB[BUCKET_B(c0, c1)];
        }
    }
    PAb = n - m;
    ISAb = m;
    for (i = m - 2; 0 <= i; --i) {
        t = SA[PAb + i];
        c0 = T(t);
        c1 = T(t + 1);
        SA[--bucketB[BUCKET_BSTAR(c0, c1)]] = i;
    }
    t = SA[PAb + m - 1];
    c0 = T(t);
    c1 = T(t + 1);
    SA[--bucketB[BUCKET_BSTAR(c0, c1)]] = m - 1;

    bufp = SA;
    bufoffset = m;
    bufsize = n - (2 * m);
    if (bufsize <= 256) {
        bufp = tempbuf;
        bufoffset = 0;
        bufsize = 256;
    }
    for (c0 = 255, j = m; 0 < j; --c0) {
        for (c1 = 255; c0 < c1; j = i, --c1) {
            i = bucketB[BUCKET_BSTAR(c0, c1)];
            if (1 < (j - i)) subStringSort(D, PAb, i, j, bufp, bufoffset, bufsize, 2, SA[i] == (m - 1), n);
        }
    }
    for (i = m - 1; 0 <= i; --i) {
        if (0 <= SA[i]) {
            j = i;
            do { SA[ISAb + SA[i]] = i; } while ((0 <= --i) && (0 <= SA[i]));
            SA[i + 1] = i - j;
            if (i <= 0) break;
        }
        j = i;
        do { SA[ISAb + (SA[i] = ~SA[i])] = j; } while (SA[--i] < 0);
        SA[ISAb + SA[i]] = j;
    }
    trSort(D, ISAb, m, 1);

    i = n - 1;
    j = m;
    if ((T(i) < T(0)) || ((T(i) == T(0)) && (flag != 0))) {
        if (flag == 0) SA[SA[ISAb + --j]] = i;
        for (--i; (0 <= i) && (T(i) <= T(i + 1)); --i) {}
    }
    for (; 0 <= i;) {
        for (--i; (0 <= i) && (T(i) >= T(i + 1)); --i) {}
        if (0 <= i) {
            SA[SA[ISAb + --j]] = i;
            for (--i; (0 <= i) && (T(i) <= T(i + 1)); --i) {}
        }
    }
    for (c0 = 255, i = n - 1, k = m - 1; 0 <= c0; --c0) {
        for (c1 = 255; c0 < c1; --c1) {
            t = i - bucketB[BUCKET_B(c0, c1)];
            bucketB[BUCKET_B(c0, c1)] = i + 1;
            for (i = t, j = bucketB[BUCKET_BSTAR(c0, c1)]; j <= k; --i, --k) SA[i] = SA[k];
        }
        t = i - bucketB[BUCKET_B(c0, c0)];
        bucketB[BUCKET_B(c0, c0)] = i + 1;
        if (c0 < 255) bucketB[BUCKET_BSTAR(c0, c0 + 1)] = t + 1;
        i = bucketA[c0];
    }
    return m;
}

/* D:3369-3481 */
static __device__ int constructBWT(dss_t *D, int *bucketA, int *bucketB) {
    const int n = D->n;
    int i, j, t = 0, s, s1, c0, c1, c2 = 0, orig = -1;
    for (c1 = 254; 0 <= c1; --c1) {
        for (i = bucketB[BUCKET_BSTAR(c1, c1 + 1)], j = bucketA[c1 + 1], t = 0, c2 = -1; i <= j; --j) {
            if (0 <= (s1 = s = SA[j])) {
                if (--s < 0) s = n - 1;
                if ((c0 = T(s)) <= c1) {
                    SA[j] = ~s1;
                    if ((0 < s) && (T(s - 1) > c0)) s = ~s;
                    if (c2 == c0) {
                        SA[--t] = s;
                    } else {
                        if (0 <= c2) bucketB[BUCKET_B(c2, c1)] = t;
                        SA[t = bucketB[BUCKET_B(c2 = c0, c1)] - 1] = s;
                    }
                }
            } else {
                SA[j] = ~s;
            }
        }
    }
    for (i = 0; i < n; ++i) {
        if (0 <= (s1 = s = SA[i])) {
            if (--s < 0) s = n - 1;
            if ((c0 = T(s)) >= T(s + 1)) {
                if ((0 < s) && (T(s - 1) < c0)) s = ~s;
                if (c0 == c2) {
                    SA[++t] = s;
                } else {
                    if (c2 != -1) bucketA[c2] = t; /* reference BUGFIX :3456 */
                    SA[t = bucketA[c2 = c0] + 1] = s;
                }
            }
        } else {
            s1 = ~s1;
        }
        if (s1 == 0) {
            SA[i] = T(n - 1);
            orig = i;
        } else {
            SA[i] = T(s1 - 1);
        }
    }
    return orig;
}


#undef SA
#undef T

// ------------------------------------------------------------------ which blocks need the replay
// The replay's result is a pure function of (period word w, k, declared block size): identical
// periodic blocks -- every full block of "ab...", of zeros, of any repeated unit -- need it once.
// Two independent 64-bit sums over the period word (position-mixed, order-independent so that a
// wave can add them in any order) plus the first 64 bytes of w itself identify w (a word of up to
// 64 bytes -- "ab", zeros, "abc", every short unit -- is compared exactly, no hash involved); the
// context caches origin pointers by (level, n, k, hashes, head of w) and a batch runs one replay per
// distinct key it has not seen.  A reused pointer is never trusted: k2_tie_verify checks every
// periodic block of the batch against the parallel sort's rank of rotation 0 (run_tie_finish).
constexpr u32 TIE_REC = 10; // u64 words per block: h1, h2, w[0..64)
__device__ __forceinline__ u64 mix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k2_tie_hash(const u32 *__restrict__ nArr,
                                                   const u32 *__restrict__ periodic,
                                                   const u8 *__restrict__ Tsrc, u64 *__restrict__ hash, u32 S) {
    const u32 b = blockIdx.x;
    const u32 n = nArr[b], k = periodic[b];
    if (k <= 1 || k >= n) return;
    const u32 p = n / k;
    const u8 *t = Tsrc + (size_t)b * S;
    u64 h1 = 0, h2 = 0;
    for (u32 i = threadIdx.x; i < p; i += 256) {
        const u64 v = (u64)t[i] + 1u;
        h1 += mix64(((u64)i << 9) | v);
        h2 += mix64((((u64)i << 9) | v) ^ 0xA5A5A5A5A5A5A5A5ull) * (2 * (u64)i + 1);
    }
    __shared__ u64 s1[256], s2[256];
    s1[threadIdx.x] = h1;
    s2[threadIdx.x] = h2;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            s1[threadIdx.x] += s1[threadIdx.x + d];
            s2[threadIdx.x] += s2[threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        hash[TIE_REC * b] = s1[0];
        hash[TIE_REC * b + 1] = s2[0];
    }
    if (threadIdx.x < 8) { // the first 64 bytes of w, zero-padded (exact identity for |w| <= 64)
        u64 v = 0;
        for (u32 j = 0; j < 8; j++) {
            const u32 i = threadIdx.x * 8 + j;
            if (i < p) v |= (u64)t[i] << (8 * j);
        }
        hash[TIE_REC * b + 2 + threadIdx.x] = v;
    }
}

// Every periodic block of the batch, whatever the source of its pointer (replay, cache, twin): the
// pointer must name one of the k rotations equal to rotation 0, which occupy the sorted positions
// [rank(0), rank(0) + k) of the parallel sort.  flag[b] = 1 and ctlw[2]++ where it does not.
__global__ __launch_bounds__(256) void k2_tie_verify(const u32 *__restrict__ nArr,
                                                     const u32 *__restrict__ periodic,
                                                     const u32 *__restrict__ rank,
                                                     const u32 *__restrict__ orig, u32 *__restrict__ flag,
                                                     u32 *__restrict__ ctlw, u32 S, u32 nb) {
    const u32 b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    const u32 n = nArr[b], k = periodic[b];
    u32 bad = 0;
    if (k > 1 && k < n) {
        const u32 r0 = rank[(size_t)b * S], o = orig[b];
        bad = (o < r0 || o >= r0 + k) ? 1u : 0u;
    }
    flag[b] = bad;
    if (bad) atomicAdd(&ctlw[2], 1u);
}

// One workgroup (one wave) per block that is marked in `todo`.  Scratch comes from the sort buffers,
// which are dead once k2_finish has run: keyA[b] holds SA (blockSize + 1 ints) and the T copy (n + 1
// bytes), keyB[b] the two bucket tables.  ctlw[0] = traps, ctlw[1] = BWT bytes that disagree with
// the parallel sort's, ctlw[2] = origin pointers outside the class of rotation 0 (both must be 0).
__global__ __launch_bounds__(64) void k2_tie_divsufsort(const u32 *__restrict__ nArr,
                                                        const u32 *__restrict__ periodic,
                                                        const u32 *__restrict__ todo,
                                                        const u8 *__restrict__ Tsrc,
                                                        const u8 *__restrict__ bwt,
                                                        const u32 *__restrict__ rank, u64 *keyA,
                                                        u64 *keyB, u32 *__restrict__ orig,
                                                        u32 *__restrict__ ctlw, u32 S,
                                                        u32 blockSize) {
    const u32 b = blockIdx.x;
    const u32 n = nArr[b];
    const u32 k = periodic[b];
    if (!todo[b] || k <= 1 || k >= n) return; // aperiodic, all-equal (orig 0, D:3168-3174,3505), or known
    int *sa = (int *)(keyA + (size_t)b * S);
    u8 *t = (u8 *)(sa + S + 2);
    int *bucketB = (int *)(keyB + (size_t)b * S);
    int *bucketA = bucketB + BUCKET_B_SIZE;
    const u8 *src = Tsrc + (size_t)b * S;
    for (u32 i = threadIdx.x; i < n; i += 64) t[i] = src[i];
    if (threadIdx.x == 0) t[n] = src[0]; // S/BZip2BlockCompressor.swift:266
    for (u32 i = threadIdx.x; i < BUCKET_B_SIZE + BUCKET_A_SIZE; i += 64) bucketB[i] = 0;
    // the reference's bwtBlock: blockSize + 1 ints, zeroed (S/BZip2BlockCompressor.swift:318)
    for (u32 i = threadIdx.x; i <= blockSize; i += 64) sa[i] = 0;
    __syncthreads();
    __shared__ int s_orig;
    if (threadIdx.x == 0) {
        dss_t D;
        D.SA = sa;
        D.T = t;
        D.n = (int)n;
        D.sa_len = (int)blockSize + 1;
        D.trap = 0;
        int m = sortTypeBstar(&D, bucketA, bucketB); // D:3501
        s_orig = (0 < m) ? constructBWT(&D, bucketA, bucketB) : 0; // D:3502-3505
        if (D.trap) atomicAdd(&ctlw[0], 1u);
    }
    __syncthreads();
    // the sorter's own BWT bytes must agree with the parallel sort's
    u32 bad = 0;
    const u8 *mine = bwt + (size_t)b * S;
    for (u32 i = threadIdx.x; i < n; i += 64) bad += ((u32)(sa[i] & 0xff) != (u32)mine[i]);
    bad = wave_reduce_add(bad);
    if (threadIdx.x == 0) {
        orig[b] = (u32)s_orig;
        if (bad) atomicAdd(&ctlw[1], bad);
        // whatever the tie order, the pointer must name one of the k rotations equal to rotation 0:
        // they occupy sorted positions [rank(0), rank(0) + k)
        const u32 r0 = rank[(size_t)b * S];
        if ((u32)s_orig < r0 || (u32)s_orig >= r0 + k) atomicAdd(&ctlw[2], 1u);
    }
}

} // namespace

// Periodic blocks of the batch: hash their period words, look the keys up in the context's cache,
// start one replay per distinct unknown key on the tie stream (it overlaps MTF / Huffman of the
// batch: only the 24 origin-pointer bits of the block header depend on it,
// S/BZip2BlockCompressor.swift:277).  run_tie_finish makes B.orig final.
void run_tie_launch(bz2hip_ctx *c, int nb) {
    Batch &B = c->batch;
    B.tie_items.clear();
    B.tie_running = false;
    u64 *d_hash = (u64 *)B.pvote;                 // votes are dead after the first doubling round
    u32 *d_todo = B.pvote + 2 * TIE_REC * (size_t)B.NB; // [NB] behind the records ([NB][TIE_REC] u64)
    k2_tie_hash<<<nb, 256, 0, c->stream>>>(B.n, B.periodic, B.T, d_hash, B.S);
    KCHECK();
    std::vector<u32> hn(nb), hk(nb);
    std::vector<u64> hh((size_t)nb * TIE_REC);
    HIPCHK(hipMemcpyAsync(hn.data(), B.n, nb * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hk.data(), B.periodic, nb * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hh.data(), d_hash, (size_t)nb * TIE_REC * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<u32> todo(nb, 0);
    std::map<TieKey, int> first_in_batch; // key -> block that runs the replay
    bool any = false;
    for (int b = 0; b < nb; b++) {
        if (hk[b] <= 1 || hk[b] >= hn[b]) continue;
        TieItem it;
        it.block = b;
        const u64 *rec = &hh[TIE_REC * (size_t)b];
        it.key = TieKey{(u32)B.level, hn[b], hk[b], rec[0], rec[1], {rec[2], rec[3], rec[4], rec[5], rec[6], rec[7], rec[8], rec[9]}};
        auto hit = c->tie_cache.find(it.key);
        if (hit != c->tie_cache.end()) {
            it.rep = -1;
            it.orig = hit->second;
            c->st_tie_cached++;
        } else {
            auto f = first_in_batch.find(it.key);
            if (f == first_in_batch.end()) {
                first_in_batch[it.key] = b;
                it.rep = b;
                todo[b] = 1;
                any = true;
                c->st_tie_replays++;
            } else {
                it.rep = f->second;
                c->st_tie_cached++;
            }
            it.orig = 0;
        }
        B.tie_items.push_back(it);
    }
    if (!any) return;
    HIPCHK(hipMemcpyAsync(d_todo, todo.data(), nb * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(B.ctl + 13, 0, 3 * sizeof(u32), c->stream));
    HIPCHK(hipEventRecord(c->ev_tie, c->stream));
    HIPCHK(hipStreamWaitEvent(c->stream_tie, c->ev_tie, 0));
    {
        hipStream_t keep = c->stream;
        c->stream = c->stream_tie; // the profiling events of this scope belong to the tie stream
        {
            ProfScope ps(c, P_BWT_TIE);
            k2_tie_divsufsort<<<nb, 64, 0, c->stream_tie>>>(B.n, B.periodic, d_todo, B.T, B.bwt, B.rank, B.keyA,
                                                            B.keyB, B.orig, B.ctl + 13, B.S,
                                                            (u32)B.level * 100000u);
        }
        c->stream = keep;
        KCHECK();
    }
    B.tie_running = true;
}

// Waits for the batch's replays, checks them, remembers their results and writes the final origin
// pointers of all periodic blocks of the batch into B.orig (on the main stream).
void run_tie_finish(bz2hip_ctx *c, int nb) {
    Batch &B = c->batch;
    if (B.tie_items.empty()) return;
    std::vector<u32> ho(nb, 0);
    if (B.tie_running) {
        HIPCHK(hipStreamSynchronize(c->stream_tie));
        u32 h[3];
        HIPCHK(hipMemcpyAsync(h, B.ctl + 13, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(ho.data(), B.orig, nb * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        B.tie_running = false;
        if (h[1])
            throw HipError(BZ2HIP_E_HIP, "internal: periodic-block sorter and parallel BWT disagree on " +
                                             std::to_string(h[1]) + " bytes");
        if (h[2])
            throw HipError(BZ2HIP_E_HIP, "internal: origin pointer of a periodic block outside the class of rotation 0");
        c->st_tie_traps += h[0];
    }
    if (c->tie_cache.size() > 8192) c->tie_cache.clear();
    for (TieItem &it : B.tie_items)
        if (it.rep == it.block) {
            it.orig = ho[it.block];
            c->tie_cache[it.key] = c->tie_poison ? it.orig + it.key.k : it.orig; // poison: tests only
        }
    for (TieItem &it : B.tie_items)
        if (it.rep >= 0 && it.rep != it.block) it.orig = ho[it.rep];
    // patch: the items are few, one small copy each
    for (const TieItem &it : B.tie_items)
        if (it.rep != it.block)
            HIPCHK(hipMemcpyAsync(B.orig + it.block, &it.orig, sizeof(u32), hipMemcpyHostToDevice, c->stream));
    // No pointer goes into a block header unchecked: cache hits and twins included (a colliding key
    // would otherwise emit a stream that decodes to a rotation of the block, silently).
    u32 *d_flag = B.pvote + 2 * TIE_REC * (size_t)B.NB;
    HIPCHK(hipMemsetAsync(B.ctl + 13, 0, 3 * sizeof(u32), c->stream));
    k2_tie_verify<<<(nb + 255) / 256, 256, 0, c->stream>>>(B.n, B.periodic, B.rank, B.orig, d_flag, B.ctl + 13, B.S, (u32)nb);
    KCHECK();
    u32 h[3];
    HIPCHK(hipMemcpyAsync(h, B.ctl + 13, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream)); // also: the patch sources above are host temporaries
    if (h[2]) {
        // a reused pointer is outside the class of rotation 0: forget those keys and replay the
        // blocks themselves (d_flag is the todo list; the sort buffers are still dead, rank alive)
        std::vector<u32> flag(nb);
        HIPCHK(hipMemcpy(flag.data(), d_flag, nb * sizeof(u32), hipMemcpyDeviceToHost));
        for (const TieItem &it : B.tie_items)
            if (flag[it.block]) {
                c->tie_cache.erase(it.key);
                c->st_tie_rejected++;
                c->st_tie_replays++;
            }
        HIPCHK(hipMemsetAsync(B.ctl + 13, 0, 3 * sizeof(u32), c->stream));
        {
            ProfScope ps(c, P_BWT_TIE);
            k2_tie_divsufsort<<<nb, 64, 0, c->stream>>>(B.n, B.periodic, d_flag, B.T, B.bwt, B.rank, B.keyA, B.keyB,
                                                        B.orig, B.ctl + 13, B.S, (u32)B.level * 100000u);
        }
        KCHECK();
        HIPCHK(hipMemcpyAsync(h, B.ctl + 13, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(ho.data(), B.orig, nb * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (h[1] || h[2])
            throw HipError(BZ2HIP_E_HIP, "internal: replay of a periodic block after a rejected cached origin pointer failed its checks");
        c->st_tie_traps += h[0];
        for (const TieItem &it : B.tie_items)
            if (flag[it.block]) c->tie_cache[it.key] = c->tie_poison ? ho[it.block] + it.key.k : ho[it.block];
    }
    B.tie_items.clear();
}
