/* Sanitizer driver for oracle/fast.c (TEST INFRASTRUCTURE): built by tests/test_oracle_sanitize.py with
 * -fsanitize=address,undefined and run once; exercises every entry point on small ragged inputs, including the edge
 * rows (index 0 and V-1), a batch of one, duplicates, and exact-size buffers so that any out-of-bounds access trips
 * AddressSanitizer.  Prints a checksum; exit code 0 = clean. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void gather_fields(const float* const* tabs, const int64_t* idx, int64_t B, int F, int E, float* out, int64_t ldo);
void scatter_fields(float* const* gtabs, const int64_t* idx, int64_t B, int F, int E, const float* d, int64_t ldd);
void adam_dense(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int step);
void adagrad_dense(float* p, const float* g, float* s, int64_t n, float lr, float eps);

static uint32_t rng = 12345u;
static uint32_t next(void) { rng = rng * 1664525u + 1013904223u; return rng >> 8; }
static float frand(void) { return (float)(next() & 0xffff) / 32768.0f - 1.0f; }

int main(void) {
  double sum = 0.0;
  const int Fs[] = {1, 3, 30}, Es[] = {1, 8, 16};
  const int64_t Bs[] = {1, 7, 257};
  for (int fi = 0; fi < 3; ++fi) for (int ei = 0; ei < 3; ++ei) for (int bi = 0; bi < 3; ++bi) {
    const int F = Fs[fi], E = Es[ei];
    const int64_t B = Bs[bi];
    float** tabs = malloc(sizeof(float*) * F);
    float** grads = malloc(sizeof(float*) * F);
    int64_t* V = malloc(sizeof(int64_t) * F);
    for (int f = 0; f < F; ++f) {
      V[f] = 1 + (int64_t)(next() % 50);
      tabs[f] = malloc(sizeof(float) * V[f] * E);   /* exact size: a read past row V-1 is a heap overflow */
      grads[f] = calloc(V[f] * E, sizeof(float));
      for (int64_t i = 0; i < V[f] * E; ++i) tabs[f][i] = frand();
    }
    int64_t* idx = malloc(sizeof(int64_t) * B * F);
    for (int64_t b = 0; b < B; ++b)
      for (int f = 0; f < F; ++f)
        idx[b * F + f] = (b == 0) ? 0 : (b == 1 ? V[f] - 1 : (int64_t)(next() % V[f]));  /* edges + duplicates */
    const int64_t ldo = (int64_t)F * E;  /* exact pitch */
    float* out = malloc(sizeof(float) * B * ldo);
    gather_fields((const float* const*)tabs, idx, B, F, E, out, ldo);
    scatter_fields(grads, idx, B, F, E, out, ldo);
    for (int f = 0; f < F; ++f) {
      const int64_t n = V[f] * E;
      float* m = calloc(n, sizeof(float));
      float* v = calloc(n, sizeof(float));
      adam_dense(tabs[f], grads[f], m, v, n, 0.005f, 0.9f, 0.999f, 1e-8f, 1);
      adam_dense(tabs[f], grads[f], m, v, n, 0.005f, 0.9f, 0.999f, 1e-8f, 2);
      adagrad_dense(tabs[f], grads[f], v, n, 0.01f, 1e-10f);
      for (int64_t i = 0; i < n; ++i) sum += tabs[f][i];
      free(m); free(v);
    }
    adam_dense(tabs[0], grads[0], grads[0], grads[0], 0, 0.1f, 0.9f, 0.999f, 1e-8f, 1);  /* empty tensor */
    for (int f = 0; f < F; ++f) { free(tabs[f]); free(grads[f]); }
    free(tabs); free(grads); free(V); free(idx); free(out);
  }
  printf("oracle/fast.c sanitizer self-test checksum %.6f\n", sum);
  return sum == sum ? 0 : 1;  /* NaN would be a bug too */
}
