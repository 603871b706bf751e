// tests/refdrop: definitions of the renderer / player members that the reference's terrain.cpp and chunk.hpp reference but whose own
// translation units (drawable.cpp, shaderProgram.cpp, player.cpp: GL / GLFW calls) are outside the generation path.  None of them
// computes anything that reaches a block id or a vertex.
#include "terrain/terrain.hpp"
#include "rendering/shaderProgram.hpp"

Drawable::Drawable() : bufIdx(0), bufVerts(0), bufFullscreenTriInfo(0) {}
Drawable::~Drawable() {}
void Drawable::destroyVBOs() { idxCount = -1; }
GLenum Drawable::drawMode() const { return 0x0004; /* GL_TRIANGLES */ }
int Drawable::getIdxCount() const { return idxCount; }

void Chunk::bufferVBOs() {}                                  // chunk.cu:2005-2021: GL upload of idx / verts

void ShaderProgram::setModelMat(const glm::mat4&) const {}
void ShaderProgram::draw(Drawable&) const {}

vec3 Player::getPos() const { return pos; }
vec3 Player::getForward() const { return forward; }
