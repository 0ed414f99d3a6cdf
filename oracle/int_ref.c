/*
 * ORACLE (test infrastructure only -- never linked into the product library).
 *
 * Plain-C restatement of the integer parts of the reference's S2T hot path.
 * Each function cites the reference lines it follows.  Built by
 * __graft_entry__.build() into oracle/_build/liboracle_int.so and loaded by
 * oracle/int_ref.py through ctypes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* First index of the maximum of a float row (ties -> lowest index).
 * Reference: `prob_ctc[b][:len].argmax(-1)`,
 * examples/speech_recognition/models/conv_transformer.py:284 (torch.argmax
 * returns the first maximal index on CPU). NaN handling is not needed: the
 * input is a softmax output. */
int32_t orc_argmax_first(const float *row, int32_t n) {
    int32_t best = 0;
    float bv = row[0];
    for (int32_t i = 1; i < n; ++i) {
        if (row[i] > bv) { bv = row[i]; best = i; }
    }
    return best;
}

/* Run-length compression of per-frame CTC predictions.
 * Reference: conv_transformer.py:283-287 (`groupby(predicted)` on the first
 * src_lengths[b] frames only; blanks are NOT removed; new length = #runs).
 *
 * pred      [B][T] int32 argmax ids (only the first lengths[b] entries used)
 * lengths   [B]    int64 valid frames per utterance
 * run_tok   [B][T] int32 out: token of run j          (-1 beyond new_len)
 * run_len   [B][T] int32 out: number of frames in run (0 beyond new_len)
 * run_start [B][T] int32 out: first frame of run j    (0 beyond new_len)
 * seg_id    [B][T] int32 out: run index of frame t    (-1 for t >= lengths[b])
 * new_len   [B]    int64 out
 * returns max(new_len) (the T'' that shapes the compressed tensor, :388) */
int64_t orc_ctc_rle(const int32_t *pred, const int64_t *lengths, int32_t B, int32_t T,
                    int32_t *run_tok, int32_t *run_len, int32_t *run_start,
                    int32_t *seg_id, int64_t *new_len) {
    int64_t mx = 0;
    for (int32_t b = 0; b < B; ++b) {
        const int32_t *p = pred + (size_t)b * T;
        int32_t *rt = run_tok + (size_t)b * T, *rl = run_len + (size_t)b * T;
        int32_t *rs = run_start + (size_t)b * T, *sg = seg_id + (size_t)b * T;
        for (int32_t t = 0; t < T; ++t) { rt[t] = -1; rl[t] = 0; rs[t] = 0; sg[t] = -1; }
        int64_t L = lengths[b];
        if (L > T) L = T;
        int32_t j = -1;
        for (int32_t t = 0; t < (int32_t)L; ++t) {
            if (j < 0 || p[t] != rt[j]) { ++j; rt[j] = p[t]; rs[j] = t; rl[j] = 0; }
            rl[j] += 1;
            sg[t] = j;
        }
        new_len[b] = (int64_t)(j + 1);
        if (new_len[b] > mx) mx = new_len[b];
    }
    return mx;
}

/* Greedy CTC decode of one utterance: collapse repeats, then drop blanks.
 * Reference: examples/speech_recognition/criterions/CTC_loss.py:50-58.
 * out must hold n ints; returns the decoded length. */
int32_t orc_ctc_greedy(const int32_t *pred, int32_t n, int32_t blank, int32_t *out) {
    int32_t m = 0;
    for (int32_t t = 0; t < n; ++t) {
        if (t > 0 && pred[t] == pred[t - 1]) continue;
        if (pred[t] == blank) continue;
        out[m++] = pred[t];
    }
    return m;
}

/* Edit-distance alignment error count.
 * Reference: examples/speech_recognition/utils/wer_utils.py:71-203
 * (EditDistance(time_mediated=False).align(refs=predicted, hyps=target)):
 * costs match 0, insertion 3, deletion 3, substitution 4; the diagonal move is
 * the default, insertion (left) wins only if strictly cheaper, deletion (up)
 * only if strictly cheaper than the best so far; errors = number of
 * alignment codes != match along the back-trace (CTC_loss.py:61-72).
 * `refs` is the decoded prediction and `hyps` the target, as in the caller
 * (CTC_loss.py:61-63).  Both empty -> the reference's align() returns NaN and
 * the caller would fail; we return 0 (never hit: targets carry EOS). */
int32_t orc_align_errors(const int32_t *refs, int32_t nr, const int32_t *hyps, int32_t nh) {
    if (nr == 0 && nh == 0) return 0;
    int32_t rows = nr + 1, cols = nh + 1;
    double *sc = (double *)malloc(sizeof(double) * (size_t)rows * cols);
    int32_t *bt = (int32_t *)malloc(sizeof(int32_t) * (size_t)rows * cols);
    for (int32_t i = 0; i < rows; ++i) {
        for (int32_t j = 0; j < cols; ++j) {
            size_t o = (size_t)i * cols + j;
            if (i == 0 && j == 0) { sc[o] = 0.0; bt[o] = 0; continue; }
            if (i == 0) { sc[o] = sc[o - 1] + 3.0; bt[o] = (int32_t)(o - 1); continue; }
            if (j == 0) { sc[o] = sc[o - cols] + 3.0; bt[o] = (int32_t)(o - cols); continue; }
            double best = sc[o - cols - 1] + ((refs[i - 1] == hyps[j - 1]) ? 0.0 : 4.0);
            int32_t prev = (int32_t)(o - cols - 1);
            double ins = sc[o - 1] + 3.0;
            if (ins < best) { best = ins; prev = (int32_t)(o - 1); }
            double del = sc[o - cols] + 3.0;
            if (del < best) { best = del; prev = (int32_t)(o - cols); }
            sc[o] = best; bt[o] = prev;
        }
    }
    int32_t errors = 0;
    int32_t cur = rows * cols - 1;
    while (cur != 0) {
        int32_t prev = bt[cur];
        int32_t cr = cur / cols, cc = cur % cols, pr = prev / cols, pc = prev % cols;
        if (cr - 1 == pr && cc - 1 == pc) {
            if (refs[cr - 1] != hyps[cc - 1]) errors += 1;   /* substitution */
        } else {
            errors += 1;                                      /* insertion / deletion */
        }
        cur = prev;
    }
    free(sc); free(bt);
    return errors;
}

/* Batch CTC unit-error counts (compute_ctc_uer, CTC_loss.py:31-74).
 * pred [B][T] greedy argmax ids, input_len [B], targets [B][L] (padded),
 * target_len [B].  Outputs summed errors and total target length. */
void orc_ctc_uer(const int32_t *pred, const int64_t *input_len, int32_t B, int32_t T,
                 const int64_t *targets, const int64_t *target_len, int32_t L, int32_t blank,
                 double *errors, double *total) {
    int32_t *dec = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T > 0 ? T : 1));
    int32_t *tgt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(L > 0 ? L : 1));
    double e = 0.0, n = 0.0;
    for (int32_t b = 0; b < B; ++b) {
        int32_t m = orc_ctc_greedy(pred + (size_t)b * T, (int32_t)input_len[b], blank, dec);
        int32_t tl = (int32_t)target_len[b];
        for (int32_t i = 0; i < tl; ++i) tgt[i] = (int32_t)targets[(size_t)b * L + i];
        e += (double)orc_align_errors(dec, m, tgt, tl);
        n += (double)tl;
    }
    free(dec); free(tgt);
    *errors = e; *total = n;
}

/* Stable descending argsort of frame counts (collate order).
 * Reference: examples/speech_recognition/data/collaters.py:86-88
 * (`frames_lengths.sort(descending=True)`); torch's CPU sort is stable for
 * 1-D int64 here, which is what the reference's known-answer collater test
 * (tests/speech_recognition/test_collaters.py) relies on. */
void orc_sort_desc(const int64_t *len, int32_t n, int64_t *order) {
    for (int32_t i = 0; i < n; ++i) order[i] = i;
    for (int32_t i = 1; i < n; ++i) {           /* insertion sort: stable */
        int64_t k = order[i]; int32_t j = i - 1;
        while (j >= 0 && len[order[j]] < len[k]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = k;
    }
}
