// knobs.h -- the ONE registry of the engine's environment switches.
//
// Every switch the library reads goes through knob("PSS_..."): a getenv() that only answers for names listed in
// kKnobs[] below (an unlisted name is a programming error: it answers "unset" and says so once on stderr, and
// tests/test_host.py compares the sources with this table).  None of the switches changes a RESULT -- they choose
// routes, sizes and diagnostics -- which is what makes them fuzzable: tests/tools/fuzz.py draws its switch settings
// from this table mechanically (the `fuzz` column: values any of which may be set, in any combination, on any input;
// empty = not drawn -- sizes of the host machine, paths, diagnostics).  The C ABI lists the table (pss_knob_count /
// pss_knob_info, include/pss.h), so tools need no copy of it.
#pragma once

namespace pss {

struct KnobDef {
    const char *name;
    const char *dflt;      // what an unset switch means
    const char *fuzz;      // '|'-separated values the fuzzer may set ("" = never drawn)
    const char *what;
};

extern const KnobDef kKnobs[];
extern const int kNumKnobs;

// getenv() of a registered switch (nullptr when unset -- or unregistered)
const char *knob(const char *name);

}  // namespace pss
