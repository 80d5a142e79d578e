// Host harness for ts-trajgen_amd/csrc/tsg_heap.h (compiled by tests/test_host.py with g++): replays a script of
// queue operations and prints what every pop returned, for comparison with CPython's heapq.
//   P <prio> <node>   push (prio, node)          U <node> <lp>   set the node's current log_prob (tie-break key)
//   O                 pop, prints the node
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tsg_heap.h"

int main(int argc, char** argv) {
  const int words = argc > 1 ? atoi(argv[1]) : 1 << 16;
  const int n_nodes = argc > 2 ? atoi(argv[2]) : 1 << 16;
  std::vector<uint64_t> heap((size_t)words, 0xdeadbeefdeadbeefull);
  std::vector<int32_t> nodes((size_t)n_nodes * ND_WORDS, 0);
  int n = 0;
  char op;
  while (scanf(" %c", &op) == 1) {
    if (op == 'P') {
      float prio;
      int node;
      if (scanf("%f %d", &prio, &node) != 2) return 2;
      HeapEnt e;
      e.prio = prio;
      e.node = node;
      heap_push(heap.data(), n, e, nodes.data());
    } else if (op == 'U') {
      int node;
      float lp;
      if (scanf("%d %f", &node, &lp) != 2) return 2;
      nodes[(size_t)node * ND_WORDS + ND_LP] = (int32_t)tsg_f2u(lp);
    } else if (op == 'O') {
      if (n == 0) {
        printf("-1\n");
        continue;
      }
      const HeapEnt e = heap_pop(heap.data(), n, nodes.data());
      printf("%d\n", e.node);
    }
  }
  printf("n %d\n", n);
  return 0;
}
