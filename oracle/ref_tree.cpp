// oracle/ref_tree.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver around the REAL reference classes
//   tnco::node::Node   (/root/reference/include/tnco/node.hpp:32-107)
//   tnco::tree::Tree   (/root/reference/include/tnco/tree.hpp:34-204)
// compiled from the reference sources where they lie (they only need the
// pybind11 + Python headers, both present in this image).  Output goes to
// oracle/_ref/libref_tree.so (git-ignored, travels with gpurun).  It pins
// oracle/tnco_oracle.c's orc_swap_with_nn / orc_tree_is_valid.
//
// Nothing else of the reference hot path can be built here: every other
// header reaches <boost/dynamic_bitset.hpp> through tnco/bitset.hpp:21 and
// Boost is not installed (no stand-in is written, by rule).
//
// The library is loaded with ctypes from inside a Python process, so the
// CPython symbols pybind11's inline functions may reference resolve there.
#include <cstdint>
#include <stdexcept>
#include <vector>

#include <tnco/node.hpp>
#include <tnco/tree.hpp>

using node_t = tnco::node::Node<int_fast32_t>;
using tree_t = tnco::tree::Tree<node_t>;

static std::vector<node_t> make_nodes(int32_t N, const int32_t* left, const int32_t* right,
                                      const int32_t* parent) {
  std::vector<node_t> nodes;
  nodes.reserve(N);
  for (int32_t i = 0; i < N; ++i) nodes.emplace_back(std::array<int_fast32_t, 2>{left[i], right[i]}, parent[i]);
  return nodes;
}

extern "C" {

// 1 valid, 0 invalid (constructor of Node or Tree threw / is_valid false).
int ref_tree_is_valid(int32_t N, const int32_t* left, const int32_t* right, const int32_t* parent) {
  try {
    tree_t t{make_nodes(N, left, right, parent)};
    return t.is_valid().first ? 1 : 0;
  } catch (const std::exception&) {
    return 0;
  }
}

// Applies Tree::swap_with_nn(pos) in place. Returns 0 ok, -1 if the tree was invalid.
int ref_tree_swap_with_nn(int32_t N, int32_t* left, int32_t* right, int32_t* parent, int32_t pos) {
  try {
    tree_t t{make_nodes(N, left, right, parent)};
    t.swap_with_nn(pos);
    for (int32_t i = 0; i < N; ++i) {
      left[i] = (int32_t)t.nodes[i].children[0];
      right[i] = (int32_t)t.nodes[i].children[1];
      parent[i] = (int32_t)t.nodes[i].parent;
    }
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}

int ref_tree_n_leaves(int32_t N, const int32_t* left, const int32_t* right, const int32_t* parent) {
  try {
    tree_t t{make_nodes(N, left, right, parent)};
    return (int)t.n_leaves();
  } catch (const std::exception&) {
    return -1;
  }
}
}
