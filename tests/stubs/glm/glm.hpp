// Type-only stand-in for <glm/glm.hpp> — see tests/stubs/README.md.  Declarations only: enough for
// the reference's headers (core/accelerators/*.h, core/camera/camera.h, core/engines/meshEngine.h)
// to parse under -fsyntax-only.  No arithmetic is implemented here and nothing links against it.
#pragma once
namespace glm {
struct vec2 {
    float x, y;
    vec2();
    vec2(float, float);
};
struct vec3 {
    float x, y, z;
    vec3();
    explicit vec3(float);
    vec3(float, float, float);
};
struct vec4 {
    float x, y, z, w;
    vec4();
    vec4(float, float, float, float);
};
vec3 operator+(const vec3 &, const vec3 &);
vec3 operator-(const vec3 &, const vec3 &);
vec3 operator*(const vec3 &, const vec3 &);
vec3 operator/(const vec3 &, const vec3 &);
vec3 operator*(const vec3 &, float);
}  // namespace glm
