// stand-in: pangolin::OpenGlMatrix as the reference's driver reads it (m[16], column-major) and IdentityMatrix()
#pragma once
namespace pangolin {
typedef double GLprecision;
struct OpenGlMatrix {
  GLprecision m[16];
};
inline OpenGlMatrix IdentityMatrix() {
  OpenGlMatrix r;
  for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
  return r;
}
}  // namespace pangolin
