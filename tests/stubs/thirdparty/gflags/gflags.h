// stand-in: the two gflags macros the reference's driver files use
#pragma once
#define DECLARE_bool(name) extern bool FLAGS_##name
#define DEFINE_bool(name, value, help) bool FLAGS_##name = (value)
