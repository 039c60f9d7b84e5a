/* A host that is not Python (stands for the Go server: cgo links the same C ABI) lowering a rule table through
 * libcerbos_lower.so.  usage: lower_host LIB RULETABLE.pb IMAGE.out [flags] [globals-json]
 * Lowers once on the main thread, then again on two other threads at once (the library serialises them), checks that all
 * three images are the same bytes, writes the image and prints the statistics.  Exit status = the library's status. */
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int (*lower_fn)(const uint8_t*, size_t, const char*, uint32_t, uint8_t**, size_t*, char**);
typedef void (*free_fn)(void*);
typedef char* (*stats_fn)(void);
typedef int (*abi_fn)(void);

static lower_fn g_lower;
static free_fn g_free;
static const uint8_t* g_pb;
static size_t g_len;
static const char* g_globals;
static uint32_t g_flags;

struct result { int status; uint8_t* image; size_t len; };

static void* worker(void* arg) {
  struct result* r = (struct result*)arg;
  char* err = NULL;
  r->status = g_lower(g_pb, g_len, g_globals, g_flags, &r->image, &r->len, &err);
  if (err) g_free(err);
  return NULL;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: lower_host LIB RULETABLE.pb IMAGE.out [flags] [globals-json]\n"); return 64; }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 65; }
  g_lower = (lower_fn)dlsym(h, "cbl_lower_ruletable_pb");
  g_free = (free_fn)dlsym(h, "cbl_free");
  stats_fn stats = (stats_fn)dlsym(h, "cbl_last_stats_json");
  abi_fn abi = (abi_fn)dlsym(h, "cbl_abi_version");
  if (!g_lower || !g_free || !stats || !abi || abi() != 1) { fprintf(stderr, "missing symbol or ABI mismatch\n"); return 66; }
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 67; }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* pb = (uint8_t*)malloc(n ? n : 1);
  if (fread(pb, 1, n, f) != (size_t)n) { perror("read"); return 67; }
  fclose(f);
  g_pb = pb; g_len = (size_t)n;
  g_flags = argc > 4 ? (uint32_t)strtoul(argv[4], NULL, 0) : 0;
  g_globals = argc > 5 ? argv[5] : NULL;

  uint8_t* image = NULL; size_t image_len = 0; char* err = NULL;
  int st = g_lower(g_pb, g_len, g_globals, g_flags, &image, &image_len, &err);
  if (st != 0) { fprintf(stderr, "%s\n", err ? err : "(no message)"); if (err) g_free(err); return st; }
  char* s = stats();
  printf("%s\n", s ? s : "{}");
  if (s) g_free(s);

  struct result r[2]; pthread_t t[2];
  memset(r, 0, sizeof(r));
  for (int i = 0; i < 2; ++i) pthread_create(&t[i], NULL, worker, &r[i]);
  for (int i = 0; i < 2; ++i) pthread_join(t[i], NULL);
  for (int i = 0; i < 2; ++i) {
    if (r[i].status != 0 || r[i].len != image_len || memcmp(r[i].image, image, image_len) != 0) { fprintf(stderr, "thread %d: a different result\n", i); return 68; }
    g_free(r[i].image);
  }
  f = fopen(argv[3], "wb");
  if (!f || fwrite(image, 1, image_len, f) != image_len) { perror(argv[3]); return 67; }
  fclose(f);
  g_free(image);
  free(pb);
  return 0;
}
