// format_selftest -- the integer "%.8g" formatter of host/pcd_io.hpp against snprintf on pseudo-random float bit
// patterns (all exponents) and on values around the decade and rounding boundaries.  Prints the number of
// mismatches; exit code 0 iff none.   usage: format_selftest [count]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "pcd_io.hpp"

static uint64_t mix(uint64_t v) {
  v += 0x9e3779b97f4a7c15ull;
  v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ull;
  v = (v ^ (v >> 27)) * 0x94d049bb133111ebull;
  return v ^ (v >> 31);
}

int main(int argc, char **argv) {
  const uint64_t count = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 20000000ull;
  uint64_t bad = 0, fast = 0;
  std::string a, b;
  auto check = [&](float v) {
    a.clear();
    b.clear();
    if (pcp_amd::detail::put_float_fixed(a, v)) {
      ++fast;
      pcp_amd::detail::put_float_printf(b, v);
      if (a != b) {
        if (bad < 10) std::fprintf(stderr, "mismatch: %.17g -> '%s' vs '%s'\n", static_cast<double>(v), a.c_str(), b.c_str());
        ++bad;
      }
    }
  };
  for (uint64_t i = 0; i < count; ++i) {
    const uint64_t r = mix(i);
    uint32_t bits = static_cast<uint32_t>(r);
    if ((r >> 32) & 1u) bits = (bits & 0x807fffffu) | ((100u + (bits >> 23) % 60u) << 23);  // half of them 1e-8 .. 1e10
    float v;
    std::memcpy(&v, &bits, 4);
    check(v);
  }
  // neighbours of powers of ten and of k * 10^j / 2 style ties
  for (int j = -5; j <= 9; ++j)
    for (int k = 1; k <= 2000; ++k) {
      float v = static_cast<float>(k * std::pow(10.0, j) / 16.0);
      for (int s = 0; s < 5; ++s) {
        check(v);
        check(-v);
        v = std::nextafterf(v, 1e30f);
      }
    }
  std::printf("%llu values through the integer path, %llu mismatches\n", static_cast<unsigned long long>(fast),
              static_cast<unsigned long long>(bad));
  return bad ? 1 : 0;
}
