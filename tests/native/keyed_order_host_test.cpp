// The device's counter RNG and keyed orderings (open_spiel_amd/csrc/osg_common.h: Rng, order_key, path hash,
// fill_base / fill_key incl. the two-part form the search kernel uses) against the oracle's restatement
// (oracle/spiel_oracle_core.cpp), over random inputs: the replay parity of every search, rollout and trajectory rests
// on these being the same functions.
//   hipcc --cuda-host-only -x hip -O2 -I open_spiel_amd/csrc -I oracle tests/native/keyed_order_host_test.cpp oracle/liboracle.so
#include <cstdio>
#include <cstdint>
#include "osg_common.h"
#include "spiel_oracle.h"

int main() {
  uint64_t z = 0x1234567ULL;
  auto rnd = [&]() { z += 0x9E3779B97F4A7C15ULL; return osg::mix64(z); };
  long n = 0;
  for (int it = 0; it < 200000; ++it) {
    const uint64_t seed = rnd(), root = rnd() >> (rnd() & 63), sub = rnd() >> (rnd() & 63);
    const int action = static_cast<int>(rnd() % 130), cell = static_cast<int>(rnd() % 128);
    if (osg::mix64(seed) != osg_oracle::Mix64(seed)) { printf("mix64\n"); return 1; }
    osg::Rng a(seed, root, sub);
    osg_oracle::CounterRng b(seed, root, sub);
    for (int k = 0; k < 4; ++k) {
      const uint32_t m = static_cast<uint32_t>(rnd() % 1000) + 1;
      if (a.below(m) != b.Below(m)) { printf("below\n"); return 1; }
      if (a.unit() != b.Unit()) { printf("unit\n"); return 1; }
      if (a.next() != b.Next()) { printf("next\n"); return 1; }
    }
    if (osg::path_hash_root() != osg_oracle::PathHashRoot()) { printf("path root\n"); return 1; }
    const uint64_t ph = rnd();
    if (osg::path_hash_child(ph, action) != osg_oracle::PathHashChild(ph, action)) { printf("path child\n"); return 1; }
    const uint64_t ob = osg::order_base(seed, root);
    if (ob != osg_oracle::OrderBase(seed, root)) { printf("order base\n"); return 1; }
    if (osg::order_key(ob, ph, action) != osg_oracle::OrderKey(ob, ph, action)) { printf("order key\n"); return 1; }
    const uint64_t fb = osg::fill_base(seed, root, sub);
    if (fb != osg_oracle::FillBase(seed, root, sub)) { printf("fill base\n"); return 1; }
    if (osg::fill_base_of(osg::fill_root(seed, root), sub) != fb) { printf("fill base (two parts)\n"); return 1; }
    if (osg::fill_key(fb, cell) != osg_oracle::FillKey(fb, cell)) { printf("fill key\n"); return 1; }
    ++n;
  }
  printf("ok: %ld random inputs\n", n);
  return 0;
}
