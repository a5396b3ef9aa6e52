// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C-ABI callers of the two file readers the reference VENDORS and that build with plain g++ (everything else of the reference
// needs drjit + CUDA + OptiX and cannot be built here, oracle/README.md):
//   * tinyexr + miniz, exactly as BitmapLoader::load_openexr_rgba drives them (src/core/bitmap_loader.cpp:5-8,12-52:
//     TINYEXR_USE_MINIZ 0 with the reference's own miniz.h / src/core/miniz.cpp, LoadEXR -> RGBA floats);
//   * tiny_obj_loader, exactly as Mesh::load drives it (src/shape/mesh.cpp:5-6,166-243: LoadObj with the default
//     triangulation, attrib.vertices / attrib.texcoords, per face corner idx.vertex_index / idx.texcoord_index).
// This file holds no reference code: it includes the reference's headers from where they lie (-I/root/reference/include) and is
// linked with /root/reference/src/core/miniz.cpp by oracle/Makefile's `ref` target into oracle/_ref/libref_loaders.so.
// tests/test_ref_loaders.py checks psdr_jit_amd/exr.py and the product's OBJ reader against it on every tutorial data file.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define TINYEXR_USE_MINIZ 0
#include <psdr/core/miniz.h>
#define TINYEXR_IMPLEMENTATION
#include <psdr/core/tinyexr.h>

#define TINYOBJLOADER_IMPLEMENTATION
#include <tiny_obj_loader/tiny_obj_loader.h>

extern "C" {

// -> 0 on success; *out = malloc'ed width*height*4 floats (RGBA, row-major), release with ref_free
int ref_load_exr_rgba(const char *file_name, float **out, int *width, int *height) {
    const char *err = nullptr;
    int ret = LoadEXR(out, width, height, file_name, &err);
    if (ret != TINYEXR_SUCCESS) {
        if (err) FreeEXRErrorMessage(err);
        return ret ? ret : -1;
    }
    return 0;
}

struct ref_obj {
    int n_vertices, n_texcoords, n_faces;
    float *vertices;      // [n_vertices*3]
    float *texcoords;     // [n_texcoords*2] or NULL
    int *faces;           // [n_faces*3] vertex indices
    int *face_uvs;        // [n_faces*3] texcoord indices or NULL
};

// mesh.cpp:166-243 -> 0 on success; arrays malloc'ed, release with ref_free_obj
int ref_load_obj(const char *fname, ref_obj *o) {
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string warn, err;
    if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, fname)) return -1;
    std::memset(o, 0, sizeof(*o));
    o->n_vertices = (int) attrib.vertices.size() / 3;
    o->vertices = (float *) std::malloc(sizeof(float) * attrib.vertices.size());
    std::memcpy(o->vertices, attrib.vertices.data(), sizeof(float) * attrib.vertices.size());
    const bool has_uv = !attrib.texcoords.empty();
    if (has_uv) {
        o->n_texcoords = (int) attrib.texcoords.size() / 2;
        o->texcoords = (float *) std::malloc(sizeof(float) * attrib.texcoords.size());
        std::memcpy(o->texcoords, attrib.texcoords.data(), sizeof(float) * attrib.texcoords.size());
    }
    int nf = 0;
    for (const auto &s : shapes) nf += (int) s.mesh.num_face_vertices.size();
    o->n_faces = nf;
    o->faces = (int *) std::malloc(sizeof(int) * 3 * (size_t) nf);
    if (has_uv) o->face_uvs = (int *) std::malloc(sizeof(int) * 3 * (size_t) nf);
    int k = 0;
    for (const auto &s : shapes)
        for (size_t f = 0; f < s.mesh.num_face_vertices.size(); ++f) {
            if (s.mesh.num_face_vertices[f] != 3) return -2;
            for (int i = 0; i < 3; ++i) {
                const auto idx = s.mesh.indices[3 * f + i];
                o->faces[3 * k + i] = idx.vertex_index;
                if (has_uv) o->face_uvs[3 * k + i] = idx.texcoord_index;
            }
            ++k;
        }
    return 0;
}

void ref_free(void *p) { std::free(p); }
void ref_free_obj(ref_obj *o) { std::free(o->vertices); std::free(o->texcoords); std::free(o->faces); std::free(o->face_uvs); }

}
