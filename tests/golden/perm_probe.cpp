// Authoring-container helper for make_build_golden.py: prints parlay::random_permutation<int>(n) (the
// reference builder's insertion order, vamana/index.h:233) as raw int32.  Compiled against the parlay
// headers where they lie under /root/reference; only its OUTPUT is stored as a fixture.
#include <cstdio>
#include <cstdlib>

#include "parlay/primitives.h"
#include "parlay/random.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const int n = atoi(argv[1]);
  auto p = parlay::random_permutation<int>(n);
  fwrite(p.begin(), sizeof(int), (size_t)n, stdout);
  return 0;
}
