// The engine's environment switches as one struct, read once (switches.def is the table).  Fragment of engine.hip.
#pragma once
#include <cstdlib>
#include <string>

struct Switches {
#define LQ_SWITCH_B(field) bool field;
#define LQ_SWITCH_F(field) bool field;
#define LQ_SWITCH_I(field) int field;
#define LQ_SWITCH_D(field) double field;
#define LQ_SWITCH_S(field) const char* field;
#define LQ_SWITCH(T, field, env, def, doc) LQ_SWITCH_##T(field)
#include "switches.def"
#undef LQ_SWITCH
#undef LQ_SWITCH_B
#undef LQ_SWITCH_F
#undef LQ_SWITCH_I
#undef LQ_SWITCH_D
#undef LQ_SWITCH_S
};

static const Switches& sw() {
    static const Switches s = [] {
        Switches v;
#define LQ_READ_B(field, env, def) { const char* t = getenv(env); v.field = t ? atoi(t) != 0 : (bool)(def); }
#define LQ_READ_F(field, env, def) { v.field = getenv(env) != nullptr; }
#define LQ_READ_I(field, env, def) { const char* t = getenv(env); v.field = t ? atoi(t) : (int)(def); }
#define LQ_READ_D(field, env, def) { const char* t = getenv(env); v.field = t ? atof(t) : (double)(def); }
#define LQ_READ_S(field, env, def) { const char* t = getenv(env); v.field = (t && *t) ? t : nullptr; }
#define LQ_SWITCH(T, field, env, def, doc) LQ_READ_##T(field, env, def)
#include "switches.def"
#undef LQ_SWITCH
#undef LQ_READ_B
#undef LQ_READ_F
#undef LQ_READ_I
#undef LQ_READ_D
#undef LQ_READ_S
        return v;
    }();
    return s;
}

// "NAME default -- effect" per line, generated from the same table (tools/README.md quotes it; a debugging aid, not part of the path)
extern "C" const char* lqrrt_switches_describe(void) {
    static const std::string text = [] {
        std::string t;
#define LQ_SWITCH(T, field, env, def, doc) t += std::string(env) + "  (default " #def ")  " + doc + "\n";
#include "switches.def"
#undef LQ_SWITCH
        return t;
    }();
    return text.c_str();
}
