"""Scenes built in the test, handed to BOTH sides through their flat scene descriptions:
the GPU through the C ABI (ssx_upload_scene(ssx_scene_desc)) and the CPU oracle through
orc_scene_create_custom.  TEST INFRASTRUCTURE.

A CustomScene starts from one of the reference's scenes as the host library builds it (spectra,
materials, colour tables, camera) and lets the test replace the geometry, the materials and the
camera -- enough to force the branches the built-in scenes never reach (degenerate spherical
triangles, zero-area lights, rays through shared edges, ...).  Triangle normals and the light list
are inputs of the ABI; they are taken from the oracle, which derives them as the reference does
(src/geometry.hpp:62-69, src/scene.cpp:26-30).
"""
import ctypes as C

import numpy as np

import oracle_lib as ol
from simple_spectral_amd import _capi
from simple_spectral_amd.renderer import Scene


def look_at_pv_inv(eye, target, up, vfov_deg, aspect=1.0, near=0.1, far=1.0):
    """inverse(P*V) of a right-handed perspective camera (float64; both sides receive the same
    16 doubles, so how it is derived is not part of the parity contract)."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye; f /= np.linalg.norm(f)
    s = np.cross(f, up); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    V = np.eye(4)
    V[0, :3], V[1, :3], V[2, :3] = s, u, -f
    V[:3, 3] = -V[:3, :3] @ eye
    h = 1.0 / np.tan(0.5 * np.radians(vfov_deg))
    P = np.zeros((4, 4))
    P[0, 0] = h / aspect; P[1, 1] = h
    P[2, 2] = -(far + near) / (far - near); P[2, 3] = -(2.0 * far * near) / (far - near); P[3, 2] = -1.0
    inv = np.linalg.inv(P @ V)
    return np.ascontiguousarray(inv.T.reshape(16))  # column-major m[c*4+r]


class CustomScene:
    def __init__(self, base="cornell", observer=1931, texture="test-img.png", keep_quads=True):
        self.observer = observer
        self._base = Scene(base, observer=observer, texture=texture if base != "cornell" else None)
        d = self._base.desc.contents
        self.pv_inv = np.array(d.pv_inv[:], dtype=np.float64)
        self.cam_pos = np.array(d.cam_pos[:], dtype=np.float32)
        self.spectra = []   # (data float32[n], low, high)
        for i in range(d.n_spectra):
            sp = d.spectra[i]
            self.spectra.append((np.array(d.samples[sp.offset:sp.offset + sp.n], dtype=np.float32), float(sp.low), float(sp.high)))
        self.materials = [dict(kind=m.kind, albedo_mode=m.albedo_mode, albedo_spectrum=m.albedo_spectrum,
                               albedo_texture=m.albedo_texture, emission_spectrum=m.emission_spectrum)
                          for m in (d.materials[i] for i in range(d.n_materials))]
        self.textures = []
        for i in range(d.n_textures):
            t = d.textures[i]
            self.textures.append(np.ctypeslib.as_array(t.rgb, shape=(t.height, t.width, 3)).copy())
        self.cam_dir = np.array(d.cam_dir[:], dtype=np.float32)
        self.kinds = {}     # quad index -> "tri": the primitive is a PrimTri of its first three vertices (src/geometry.hpp:55-74)
        self.quads = []     # (pos[4][3], st[4][2], material)
        if keep_quads:
            for i in range(d.n_quads):
                q = d.quads[i]
                vs = (q.v00, q.v10, q.v11, q.v01)
                self.quads.append((np.array([v.pos[:] for v in vs], dtype=np.float32), np.array([v.st[:] for v in vs], dtype=np.float32), int(q.material)))
        self._keep = []

    # ---- editing -------------------------------------------------------------------------------
    def add_spectrum(self, data, low, high):
        self.spectra.append((np.asarray(data, dtype=np.float32), float(low), float(high)))
        return len(self.spectra) - 1

    def add_material(self, kind=0, albedo_spectrum=None, albedo_texture=None, emission_spectrum=None):
        """emission_spectrum None -> the all-zero table of the base scene's first non-emissive material."""
        if emission_spectrum is None:
            emission_spectrum = next(m["emission_spectrum"] for m in self.materials if not self.spectra[m["emission_spectrum"]][0].any())
        m = dict(kind=kind, albedo_mode=1 if albedo_texture is not None else 0, albedo_spectrum=albedo_spectrum or 0,
                 albedo_texture=albedo_texture or 0, emission_spectrum=emission_spectrum)
        self.materials.append(m)
        return len(self.materials) - 1

    def add_quad(self, v00, v10, v11, v01, material, st=((0, 0), (1, 0), (1, 1), (0, 1))):
        self.quads.append((np.array([v00, v10, v11, v01], dtype=np.float32), np.array(st, dtype=np.float32), int(material)))
        return len(self.quads) - 1

    def add_tri(self, v0, v1, v2, material, st=((0, 0), (1, 0), (1, 1))):
        """PrimTri(material, v0, v1, v2): carried as a quad record whose fourth vertex is unused (= v0)."""
        i = self.add_quad(v0, v1, v2, v0, material, st=tuple(st) + (st[0],))
        self.kinds[i] = "tri"
        return i

    def set_camera(self, eye, target, up=(0, 1, 0), vfov_deg=40.0, aspect=1.0):
        self.pv_inv = look_at_pv_inv(eye, target, up, vfov_deg, aspect)
        self.cam_pos = np.asarray(eye, dtype=np.float32)
        f = np.asarray(target, dtype=np.float64) - np.asarray(eye, dtype=np.float64)
        self.cam_dir = (f / np.linalg.norm(f)).astype(np.float32)   # camera.dir (both sides receive these three floats)

    # ---- the two sides -------------------------------------------------------------------------
    def oracle(self):
        """ol.Oracle over orc_scene_create_custom of this description."""
        def make(lib, color):
            keep = []
            sp = (ol.SpectrumIn * len(self.spectra))()
            for i, (data, low, high) in enumerate(self.spectra):
                keep.append(np.ascontiguousarray(data, dtype=np.float32))
                sp[i].n = len(data); sp[i].low = low; sp[i].high = high
                sp[i].data = keep[-1].ctypes.data_as(C.POINTER(C.c_float))
            mt = (ol.MaterialIn * len(self.materials))()
            for i, m in enumerate(self.materials):
                mt[i].kind = m["kind"]; mt[i].albedo_mode = m["albedo_mode"]; mt[i].albedo_spectrum = m["albedo_spectrum"]
                mt[i].texture = m["albedo_texture"]; mt[i].emission_spectrum = m["emission_spectrum"]
            tx = (ol.TextureIn * max(1, len(self.textures)))()
            for i, t in enumerate(self.textures):
                keep.append(np.ascontiguousarray(t, dtype=np.uint8))
                tx[i].w = t.shape[1]; tx[i].h = t.shape[0]; tx[i].rgb = keep[-1].ctypes.data_as(C.POINTER(C.c_uint8))
            qs = (ol.QuadIn * len(self.quads))()
            for i, (pos, st, m) in enumerate(self.quads):
                for v in range(4):
                    for k in range(3):
                        qs[i].pos[v][k] = float(pos[v][k])
                    for k in range(2):
                        qs[i].st[v][k] = float(st[v][k])
                qs[i].material = m
                qs[i].kind = 1 if self.kinds.get(i) == "tri" else 0
            pv = (C.c_double * 16)(*[float(x) for x in self.pv_inv])
            cp = (C.c_float * 3)(*[float(x) for x in self.cam_pos])
            sc = lib.orc_scene_create_custom(color, pv, cp, sp, len(self.spectra), mt, len(self.materials), tx, len(self.textures), qs, len(self.quads))
            if sc:
                lib.orc_scene_set_camera_dir(sc, (C.c_float * 3)(*[float(x) for x in self.cam_dir]))
            return sc
        return ol.Oracle(observer=self.observer, custom=make)

    def desc(self, orc):
        """ssx_scene_desc of this description (normals and light list as the oracle `orc` derived them)."""
        base = self._base.desc.contents
        d = _capi.SsxSceneDesc.from_buffer_copy(base)
        keep = self._keep = []
        for k in range(16):
            d.pv_inv[k] = float(self.pv_inv[k])
        for k in range(3):
            d.cam_pos[k] = float(self.cam_pos[k])
            d.cam_dir[k] = float(self.cam_dir[k])
        n_samples = sum(len(s[0]) for s in self.spectra)
        samples = (C.c_float * n_samples)()
        spectra = (_capi.SsxSpectrum * len(self.spectra))()
        off = 0
        for i, (data, low, high) in enumerate(self.spectra):
            samples[off:off + len(data)] = [float(x) for x in data]
            spectra[i].offset = off; spectra[i].n = len(data); spectra[i].low = low; spectra[i].high = high
            # float(n-1)/(high-low) in float arithmetic (src/spectrum.cpp:22-25)
            spectra[i].delta_recip = float(np.float32(len(data) - 1) / (np.float32(high) - np.float32(low)))
            off += len(data)
        mats = (_capi.SsxMaterial * len(self.materials))()
        for i, m in enumerate(self.materials):
            mats[i].kind = m["kind"]; mats[i].albedo_mode = m["albedo_mode"]; mats[i].albedo_spectrum = m["albedo_spectrum"]
            mats[i].albedo_texture = m["albedo_texture"]; mats[i].emission_spectrum = m["emission_spectrum"]
        texs = (_capi.SsxTexture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            keep.append(np.ascontiguousarray(t, dtype=np.uint8))
            texs[i].width = t.shape[1]; texs[i].height = t.shape[0]; texs[i].rgb = keep[-1].ctypes.data_as(C.POINTER(C.c_uint8))
        quads = (_capi.SsxQuad * len(self.quads))()
        nl = C.c_int()
        orc.lib.orc_scene_counts(orc.scene, None, C.byref(nl), None)
        lights = (C.c_uint32 * nl.value)(*[orc.lib.orc_scene_light(orc.scene, i) for i in range(nl.value)])
        n6 = (C.c_float * 6)()
        for i, (pos, st, m) in enumerate(self.quads):
            for v, name in enumerate(("v00", "v10", "v11", "v01")):
                vert = getattr(quads[i], name)
                for k in range(3):
                    vert.pos[k] = float(pos[v][k])
                for k in range(2):
                    vert.st[k] = float(st[v][k])
            orc.lib.orc_scene_quad_normals(orc.scene, i, n6)
            for k in range(3):
                quads[i].normal0[k] = n6[k]; quads[i].normal1[k] = n6[3 + k]
            quads[i].material = m
            quads[i].flags = (_capi.SSX_PRIM_LIGHT if i in list(lights) else 0) | (_capi.SSX_PRIM_TRI if self.kinds.get(i) == "tri" else 0)
        d.spectra = C.cast(spectra, C.POINTER(_capi.SsxSpectrum)); d.n_spectra = len(self.spectra)
        d.samples = C.cast(samples, C.POINTER(C.c_float)); d.n_samples = n_samples
        d.materials = C.cast(mats, C.POINTER(_capi.SsxMaterial)); d.n_materials = len(self.materials)
        d.quads = C.cast(quads, C.POINTER(_capi.SsxQuad)); d.n_quads = len(self.quads)
        d.lights = C.cast(lights, C.POINTER(C.c_uint32)); d.n_lights = nl.value
        d.textures = C.cast(texs, C.POINTER(_capi.SsxTexture)); d.n_textures = len(self.textures)
        keep += [samples, spectra, mats, texs, quads, lights]
        return d
