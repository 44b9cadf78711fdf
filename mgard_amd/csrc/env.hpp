// MGH_* developer switches (cross-checks and experiments; DESIGN.md lists them). They are read when
// a hierarchy / a high-level call is set up, and their values are VALIDATED: a value outside the
// range of its switch is an error, never a silent default. An MGH_* name the library does not know
// gets one warning on stderr per process.
#pragma once
#include <cerrno>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

extern char **environ;

namespace mgh {

struct EnvSwitch {
  const char *name;
  long lo, hi;  // accepted range (MGH_RCH: every one of its three comma-separated values)
};

inline const EnvSwitch *env_switches(size_t *count) {
  static const EnvSwitch k[] = {
      {"MGH_FORCE_V1", 0, 1},       {"MGH_FORCE_ND", 0, 1},        {"MGH_IPK_STREAM", 0, 1},      {"MGH_IPK_DMA", 0, 1},  {"MGH_IPK_DMA_MIN", 0, 1 << 30},  {"MGH_MULTI_FORCE_PEER", 0, 1},  {"MGH_ABSMAX_WARM_MB", 0, 1 << 20},      {"MGH_FUSED_FACES", 0, 1},
      {"MGH_FUSED_XCD", 0, 2},      {"MGH_FUSED_FIXED", 0, 1},     {"MGH_FUSED_WIDE", 0, 2},
      {"MGH_SLICE_BATCH", 0, 1},      {"MGH_FUSED_TALL", 0, 1},
      {"MGH_FUSED4", 0, 1},         {"MGH_CLS1", 0, 1 << 30},     {"MGH_CLS2", 0, 1 << 30},      {"MGH_RCH", 1, 16},
      {"MGH_IPK_W", 16, 64},        {"MGH_IPK_PD", 1, 4},          {"MGH_NO_RECOMPOSE_HEAD", 0, 1}, {"MGH_RESTORE_ROWS", 0, 1}, {"MGH_DEBUG_SYNC", 0, 1},
      {"MGH_HL_TIMING", 0, 1},      {"MGH_HUFF_TB", 8, 15},        {"MGH_HUFF_SERIAL_DECODE", 0, 1},
      {"MGH_HUFF_PAR_DECODE", 0, 1}, {"MGH_SYM16_DECODE", 0, 1},   {"MGH_BOX", 0, 3},            {"MGH_IPK_WPC", 1, 16},
      {"MGH_TAIL_SOLVES", 0, 1},    {"MGH_IPK_CONTIG", 2, 1 << 20},
      {"MGH_RESTORE_V", 2, 3},
      {"MGH_IPK_KR16", 0, 1},
      {"MGH_IPK_RANGE_MB", 0, 1 << 20},
      {"MGH_HL_PIPELINE", 0, 1},
      {"MGH_SYM16_MIXED", 0, 1},
      {"MGH_IPK_SPEC", 0, 1},
      {"MGH_IPK_SPEC_K", 0, 4096},
      {"MGH_IPK_SPEC_LONG", 0, 1 << 30},
      {"MGH_IPK_SPEC_MAX", 0, 1 << 30},
      {"MGH_IPK_CHUNK", 0, 1},
      {"MGH_OUTLIER_AGG", 0, 2},
      {"MGH_ND_IPK", 0, 1},
      {"MGH_HUFF_SYNC", 0, 1},
      {"MGH_HUFF_SYNC_DECODE", 0, 1},
      {"MGH_IPK_CHUNK_K", 0, 64},
      {"MGH_HL_STREAM_NORM", 0, 1},
      {"MGH_ND_ROWS", 0, 1},
      {"MGH_HUFF_PAIR", 0, 2},
      {"MGH_HL_COPY_THREADS", 1, 32},
      {"MGH_HL_RING_MB", 1, 256},
      {"MGH_HL_DECODE_FOLLOWS", 0, 1},
      {"MGH_HL_COPY_AFFINITY", 0, 1},
      {"MGH_IPK_DMA_ROUNDS", 1, 64},
      {"MGH_HL_COPY_PARTS", 1, 64},
      {"MGH_HUFF_LEAN", 0, 1},
      {"MGH_INLINE_QP", 0, 1},
      {"MGH_HUFF_DBG", 0, 7},
  };
  *count = sizeof(k) / sizeof(k[0]);
  return k;
}

inline bool env_parse_long(const char *txt, long &v) {
  if (!txt || !*txt) return false;
  char *end = nullptr;
  errno = 0;
  v = std::strtol(txt, &end, 10);
  return errno == 0 && end && *end == '\0';
}

// MGH_* names this library does not know (a typo, or another product's variable): ONE warning on
// stderr per process, never an error -- an unrelated MGH_-prefixed variable in a production
// environment must not break every call. The walk over `environ` happens once.
inline void env_warn_unknown_once() {
  static std::once_flag once;
  std::call_once(once, [] {
    size_t n = 0;
    const EnvSwitch *k = env_switches(&n);
    for (char **e = environ; e && *e; e++) {
      if (std::strncmp(*e, "MGH_", 4) != 0) continue;
      const char *eq = std::strchr(*e, '=');
      if (!eq) continue;
      const std::string name(*e, eq - *e);
      bool known = false;
      for (size_t i = 0; i < n; i++) known |= name == k[i].name;
      if (!known)
        std::fprintf(stderr, "libmgard_hip: unknown developer switch %s ignored (DESIGN.md lists them)\n",
                     name.c_str());
    }
  });
}

// Checks the value of every KNOWN switch that is set (one getenv per switch, no walk over the
// process environment); empty string = fine. A value outside the range of its switch is an error,
// never a silent default.
inline std::string env_validate() {
  env_warn_unknown_once();
  size_t n = 0;
  const EnvSwitch *k = env_switches(&n);
  for (size_t i = 0; i < n; i++) {
    const EnvSwitch *sw = &k[i];
    const char *txt = std::getenv(sw->name);
    if (!txt) continue;
    const std::string name(sw->name), val(txt);
    if (name == "MGH_RCH") {
      int parts = 0;
      size_t pos = 0;
      while (true) {
        const size_t c = val.find(',', pos);
        long v;
        if (!env_parse_long(val.substr(pos, c == std::string::npos ? c : c - pos).c_str(), v) ||
            v < sw->lo || v > sw->hi)
          return "MGH_RCH=" + val + ": expected a,b,c with every value in 1..16";
        parts++;
        if (c == std::string::npos) break;
        pos = c + 1;
      }
      if (parts != 3) return "MGH_RCH=" + val + ": expected three comma-separated values";
      continue;
    }
    long v;
    if (!env_parse_long(val.c_str(), v) || v < sw->lo || v > sw->hi)
      return name + "=" + val + ": expected an integer in " + std::to_string(sw->lo) + ".." +
             std::to_string(sw->hi);
  }
  return std::string();
}

// value of a (validated) switch, or `dflt` when it is not set
inline long env_get(const char *name, long dflt) {
  const char *e = std::getenv(name);
  long v;
  return env_parse_long(e, v) ? v : dflt;
}

} // namespace mgh
