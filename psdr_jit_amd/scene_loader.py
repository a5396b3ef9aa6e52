"""XML scene files (reference src/scene/scene_loader.cpp:174-510; a Mitsuba-0.6-style dialect): Scene.load_file /
Scene.load_string.  Host-side parsing only - the result is a sequence of the same add_Sensor / add_BSDF /
add_EnvironmentMap / add_Mesh calls a script would make, in the reference's order (sensors, BSDFs, emitters, shapes).
Supported like the reference: perspective sensors, diffuse BSDFs (colour or bitmap texture), microfacet BSDFs (constants), envmap and area emitters,
obj shapes, transforms (translate, rotate, scale, look_at, matrix).  The GGX BSDF family is not built."""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np


class _Err(RuntimeError):
    pass


def _parse_vector(text, length, allow_empty=False):
    vals = [float(t) for t in text.replace(",", " ").split()]
    if len(vals) > length:
        raise _Err("Vector too long: [%s]" % text)
    if len(vals) < length:
        if not allow_empty:
            raise _Err("Vector too short: [%s]" % text)
        vals += [vals[-1] if vals else 0.0] * (length - len(vals))
    return vals


def _child_by_name(node, names, allow_empty=False):
    for c in node:
        if c.get("name") in names:
            return c
    if not allow_empty:
        raise _Err("Missing child node: " + sorted(names)[0])
    return None


def _translate(v):
    m = np.eye(4)
    m[:3, 3] = v
    return m


def _scale(v):
    return np.diag([v[0], v[1], v[2], 1.0])


def _rotate(axis, angle_deg):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    c, s = math.cos(math.radians(angle_deg)), math.sin(math.radians(angle_deg))
    x, y, z = a
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    m = np.eye(4)
    m[:3, :3] = r
    return m


def _look_at(origin, target, up):
    """transform.h:83-104: columns left, new_up, dir, origin"""
    o, t, u = (np.asarray(v, np.float64) for v in (origin, target, up))
    d = (t - o) / np.linalg.norm(t - o)
    left = np.cross(u, d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m


def load_transform(parent):
    """scene_loader.cpp:72-126: later children are applied after (left-multiplied onto) earlier ones"""
    result = np.eye(4)
    if parent is None:
        return result.astype(np.float32)
    name = parent.get("name", "")
    if name not in ("to_world", "toWorld"):
        raise _Err("Invalid transformation name: " + name)
    for node in parent:
        tag = node.tag
        if tag == "translate":
            m = _translate([float(node.get("x", 0.0)), float(node.get("y", 0.0)), float(node.get("z", 0.0))])
        elif tag == "rotate":
            m = _rotate([float(node.get("x", 0.0)), float(node.get("y", 0.0)), float(node.get("z", 0.0))], float(node.get("angle", 0.0)))
        elif tag == "scale":
            m = _scale([float(node.get("x", 1.0)), float(node.get("y", 1.0)), float(node.get("z", 1.0))])
        elif tag in ("look_at", "lookAt", "lookat"):
            m = _look_at(_parse_vector(node.get("origin", ""), 3), _parse_vector(node.get("target", ""), 3), _parse_vector(node.get("up", ""), 3))
        elif tag == "matrix":
            m = np.asarray(_parse_vector(node.get("value", ""), 16), np.float64).reshape(4, 4)     # row-major
        else:
            raise _Err("Unsupported transformation: " + tag)
        result = m @ result
    return result.astype(np.float32)


def _load_rgb(node):
    if node.tag == "float":
        v = float(node.get("value"))
        return [v, v, v]
    if node.tag in ("rgb", "spectrum"):
        return _parse_vector(node.get("value", ""), 3, allow_empty=True)
    raise _Err("Unsupported RGB type: " + node.tag)


def _resolve(path, base_dir):
    if os.path.exists(path) or base_dir is None:
        return path
    alt = os.path.join(base_dir, path)
    return alt if os.path.exists(alt) else path


def _parse_bitmap(node, base_dir):
    if node.get("type") != "bitmap":
        raise _Err("Unsupported texture type: " + str(node.get("type")))
    fn = node.find("string")
    if fn is None or fn.get("name") != "filename" or not fn.get("value"):
        raise _Err("Failed to retrieve bitmap filename")
    return _resolve(fn.get("value"), base_dir)


def _load_plain_bsdf(node, btype, psdr, base_dir):
    """the BSDFs that can stand alone or inside a normal map (scene_loader.cpp:325-372, 381-425); None for another type"""
    b = None
    if btype == "diffuse":
        refl = _child_by_name(node, {"reflectance"})
        if refl.tag == "texture":
            b = psdr.DiffuseBSDF(psdr.Bitmap3fD(_parse_bitmap(refl, base_dir)))
        else:
            b = psdr.DiffuseBSDF(_load_rgb(refl))
    elif btype == "microfacet":
        nodes = [_child_by_name(node, {"specular_reflectance", "specularReflectance"}), _child_by_name(node, {"diffuse_reflectance", "diffuseReflectance"}),
                 _child_by_name(node, {"roughness"})]
        # load_texture (scene_loader.cpp:289-319): a <texture> child is a bitmap file, anything else a constant;
        # a one-channel Bitmap takes the first channel of the image (bitmap.cpp:38-39)
        vals = [psdr.Bitmap3fD(_parse_bitmap(n, base_dir)) if n.tag == "texture" else _load_rgb(n) for n in nodes[:2]]
        if nodes[2].tag == "texture":
            vals.append(psdr.Bitmap1fD(np.ascontiguousarray(psdr.Bitmap3fD(_parse_bitmap(nodes[2], base_dir)).data[..., 0])))
        else:
            vals.append(float(nodes[2].get("value")))
        b = psdr.MicrofacetBSDF(*vals)
    elif btype == "roughconductor":
        nodes = [_child_by_name(node, {"alpha"}), _child_by_name(node, {"eta"}), _child_by_name(node, {"k"})]
        one = lambda n: psdr.Bitmap1fD(np.ascontiguousarray(psdr.Bitmap3fD(_parse_bitmap(n, base_dir)).data[..., 0])) if n.tag == "texture" else float(n.get("value"))
        rgb = lambda n: psdr.Bitmap3fD(_parse_bitmap(n, base_dir)) if n.tag == "texture" else _load_rgb(n)
        b = psdr.RoughConductorBSDF(one(nodes[0]), rgb(nodes[1]), rgb(nodes[2]))
    elif btype == "roughdielectric":     # scene_loader.cpp:346-360
        alpha = _child_by_name(node, {"alpha"})
        ior = [_child_by_name(node, {"intIOR"}), _child_by_name(node, {"extIOR"})]
        al = psdr.Bitmap1fD(np.ascontiguousarray(psdr.Bitmap3fD(_parse_bitmap(alpha, base_dir)).data[..., 0])) if alpha.tag == "texture" else float(alpha.get("value"))
        b = psdr.RoughDielectricBSDF(al, float(ior[0].get("value")), float(ior[1].get("value")))
    return b


def load_scene(root, scene, psdr, base_dir=None):
    if root.tag != "scene":
        raise _Err("XML parsing failed")
    first_sensor = True
    for node in root.findall("sensor"):
        film, sampler = node.find("film"), node.find("sampler")
        if first_sensor:
            if film is None:
                raise _Err("Missing film node")
            if sampler is None:
                raise _Err("Missing sampler node")
            scene.opts.width = int(_child_by_name(film, {"width"}).get("value"))
            scene.opts.height = int(_child_by_name(film, {"height"}).get("value"))
            scene.opts.spp = int(sampler.find("integer").get("value"))
            scene.opts.sppe = scene.opts.sppse = 0
            first_sensor = False
        else:
            if film is not None:
                raise _Err("Duplicate film node")
            if sampler is not None:
                raise _Err("Duplicate sampler node")
        if node.get("type") != "perspective":
            raise _Err("Unsupported sensor: " + str(node.get("type")))
        to_world = load_transform(node.find("transform"))
        fov_x = float(_child_by_name(node, {"fov"}).get("value"))
        axis = _child_by_name(node, {"fov_axis", "fovAxis"}, True)
        if axis is not None and axis.get("value") != "x":
            raise _Err("Unsupported fov-axis: " + str(axis.get("value")))
        near = _child_by_name(node, {"near_clip", "nearClip"}, True)
        far = _child_by_name(node, {"far_clip", "farClip"}, True)
        cam = psdr.PerspectiveCamera(fov_x, float(near.get("value", 0.1)) if near is not None else 0.1,
                                     float(far.get("value", 1e4)) if far is not None else 1e4)
        cam.to_world = to_world
        scene.add_Sensor(cam)

    for node in root.findall("bsdf"):
        bsdf_id = node.get("id")
        if not bsdf_id:
            raise _Err("BSDF must have an id")
        btype = node.get("type")
        b = _load_plain_bsdf(node, btype, psdr, base_dir)
        if b is not None:
            pass
        elif btype == "normalmap":           # scene_loader.cpp:373-430: a <bsdf> child is the BSDF the map perturbs
            inner = node.find("bsdf")
            if inner is None:
                raise _Err("Unsupported normal map nested BSDF: none")
            nested = _load_plain_bsdf(inner, inner.get("type"), psdr, base_dir)
            if nested is None:
                raise _Err("Unsupported normal map nested BSDF: " + str(inner.get("type")))
            nm_node = _child_by_name(node, {"normalmap"})
            b = psdr.NormalMapBSDF(psdr.Bitmap3fD(_parse_bitmap(nm_node, base_dir)) if nm_node.tag == "texture" else _load_rgb(nm_node))
            b.nested_bsdf = nested
        else:
            raise _Err("Unsupported BSDF: " + str(btype))
        scene.add_BSDF(b, bsdf_id)

    for node in root.findall("emitter"):
        if node.get("type") != "envmap":
            raise _Err("Unsupported emitter: " + str(node.get("type")))
        fn = node.find("string")
        if fn is None or fn.get("name") != "filename" or not fn.get("value"):
            raise _Err("Failed to retrieve bitmap filename")
        sc_node = _child_by_name(node, {"scale"}, True)
        scale = float(sc_node.get("value", 1.0)) if sc_node is not None else 1.0
        scene.add_EnvironmentMap(_resolve(fn.get("value"), base_dir), load_transform(node.find("transform")), scale)

    for node in root.findall("shape"):
        if node.get("type") != "obj":
            raise _Err("Unsupported shape: " + str(node.get("type")))
        name_node = node.find("string")
        if name_node is None or name_node.get("name") != "filename":
            raise _Err("Missing shape filename")
        mesh = psdr.Mesh()
        mesh.load(_resolve(name_node.get("value"), base_dir))
        ref = node.find("ref")
        if ref is None:
            raise _Err("Missing BSDF reference")
        if node.find("bsdf") is not None:
            raise _Err("BSDFs declared under shapes are not supported.")
        fn_node = _child_by_name(node, {"face_normals", "faceNormals"}, True)
        mesh.use_face_normal = fn_node is not None and fn_node.get("value") == "true"
        if node.get("id"):
            mesh.id = node.get("id")
        mesh.to_world = load_transform(node.find("transform"))
        emitter = None
        em_node = node.find("emitter")
        if em_node is not None:
            if em_node.get("type") != "area":
                raise _Err("Unsupported emitter: " + str(em_node.get("type")))
            emitter = psdr.AreaLight(_load_rgb(_child_by_name(em_node, {"radiance"})))
        scene.add_Mesh(mesh, ref.get("id"), emitter)


def load_file(scene, file_name, psdr, auto_configure=True):
    try:
        root = ET.parse(file_name).getroot()
    except (ET.ParseError, OSError):
        raise RuntimeError("XML parsing failed")
    load_scene(root, scene, psdr, os.path.dirname(os.path.abspath(file_name)))
    if auto_configure:
        scene.configure()


def load_string(scene, scene_xml, psdr, auto_configure=True):
    try:
        root = ET.fromstring(scene_xml)
    except ET.ParseError:
        raise RuntimeError("XML parsing failed")
    load_scene(root, scene, psdr, None)
    if auto_configure:
        scene.configure()
