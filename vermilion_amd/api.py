"""Host-side mirror of the reference's plugin surface for the hot path.

Same names, argument meaning and error behaviour as the reference so that a
Vermilion user finds the seam unchanged:

    mEng = MeshEngine(); integrator = PathTracer(); rEng = RenderEngine(mEng)
    rEng.assignIntegrator(integrator)        # main.cpp:60-63
    mEng.loadTriangles(pos, nrm, uv)         # stands in for Assimp load + createBVH
    rEng.draw()                              # -> integrator.Render(cameras, mEng)

Reference: Integrator (core/integrators/integrators.h:11-26), Camera /
cameraSettings / pixelValue (core/camera/camera.h:21-122, camera.cpp:34-124),
RenderEngine (core/engines/renderEngine.cpp:49-166), MeshEngine::RayCast
(core/engines/meshEngine.cpp:239-509).  Scene import (Assimp), texture binding
and saveFrame (OIIO) stay with the host application and are not mirrored.
"""
import enum
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L
from .scene import Scene, make_camera, make_opts


class float3:
    """core/types/types.h float3 (only carries position/rotation into Camera)"""

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def __iter__(self):
        return iter((self.x, self.y, self.z))


class vermRenderMode(enum.Enum):  # camera.h:21-30
    RGB = 0
    RGBA = 1
    RGBAZ = 2
    Depth = 3
    Depth64 = 4
    TOTAL_RENDER_MODES = 5


_CHANNELS = {vermRenderMode.RGB: 3, vermRenderMode.RGBA: 4, vermRenderMode.RGBAZ: 5, vermRenderMode.Depth: 1}


@dataclass
class cameraSettings:  # camera.h:32-47
    imageResX: int = 2592 // 4
    imageResY: int = 1728 // 4
    position: float3 = field(default_factory=lambda: float3(-4000, 1600, 8000))
    rotation: float3 = field(default_factory=lambda: float3(0, 25, 0))
    fBackDistance: float = 6.0
    fBackSizeX: float = 3.6
    fBackSizeY: float = 2.4
    horAngleOfView: float = 90.0  # stored, unused by PathTracer (SURVEY A-16)
    raysPerPixel: int = 128
    rayMaxBounces: int = 5        # stored, unused by PathTracer
    tileSize: int = 16            # stored, unused by PathTracer
    renderMode: vermRenderMode = vermRenderMode.RGBAZ


@dataclass
class pixelValue:  # camera.h:49-58
    pixel: int = 0
    red: float = 0.0
    green: float = 0.0
    blue: float = 0.0
    alpha: float = 0.0
    depth: float = 0.0
    light: float = 0.0


class Camera:
    """Parameter + framebuffer holder (camera.cpp:34-124)."""

    def __init__(self, settings: cameraSettings):
        self.uRaysFired = 0
        self.uRaysHit = 0
        self.RenderTargetSize = settings.imageResX * settings.imageResY
        self.mDistToFilm = float(settings.fBackDistance)
        self.fAngleOfView = settings.horAngleOfView
        self.uMaxBounces = settings.rayMaxBounces
        self.mPosition = tuple(settings.position)
        # the library applies (-x,-y,+z)*3.1415926535/180 itself (camera.cpp:43-47)
        self.rotationDegrees = tuple(settings.rotation)
        self.uSamplesPerPixel = settings.raysPerPixel
        self.uTileSize = settings.tileSize
        self.uImageU = settings.imageResX
        self.uImageV = settings.imageResY
        self.renderMode = settings.renderMode
        self.sensorSizeX = float(settings.fBackSizeX)
        self.sensorSizeY = float(settings.fBackSizeY)
        if self.renderMode not in _CHANNELS:
            raise Exception("unsupported render mode")  # camera.cpp:74-80 throws std::exception
        self.mImage = np.zeros(self.RenderTargetSize * _CHANNELS[self.renderMode], np.float32)

    def setPixelValue(self, pv: pixelValue):  # camera.cpp:88-124
        m = self.renderMode
        if m == vermRenderMode.RGB:
            self.mImage[pv.pixel * 3:pv.pixel * 3 + 3] = (pv.red, pv.green, pv.blue)
        elif m == vermRenderMode.RGBA:
            self.mImage[pv.pixel * 4:pv.pixel * 4 + 4] = (pv.red, pv.green, pv.blue, pv.alpha)
        elif m == vermRenderMode.RGBAZ:
            self.mImage[pv.pixel * 5:pv.pixel * 5 + 5] = (pv.red, pv.green, pv.blue, pv.alpha, pv.depth)
        elif m == vermRenderMode.Depth:
            self.mImage[pv.pixel] = pv.depth
        else:
            raise Exception("unsupported render mode")

    def image(self):
        return self.mImage.reshape(self.uImageV, self.uImageU, _CHANNELS[self.renderMode])

    def saveFrameBuffers(self, device=0):
        """The conversion loop of Camera::saveFrame (camera.cpp:140-175) for RGBAZ, on the device:
        -> (rgba8 [H,W,4] uint8, depth [H,W] float32).  Encoding to PNG/EXR (OIIO) is the host's."""
        if self.renderMode != vermRenderMode.RGBAZ:
            raise Exception("saveFrameBuffers: RGBAZ only")
        import ctypes as C
        import torch
        dev = torch.device("cuda", device)
        frame = torch.from_numpy(self.mImage).to(dev)
        n = self.RenderTargetSize
        rgba = torch.empty((n, 4), dtype=torch.uint8, device=dev)
        depth = torch.empty((n,), dtype=torch.float32, device=dev)
        L.check(L.lib().vmx_quantize_device(C.c_void_p(frame.data_ptr()), n, C.c_void_p(rgba.data_ptr()),
                                            C.c_void_p(depth.data_ptr()), device, None))
        return (rgba.cpu().numpy().reshape(self.uImageV, self.uImageU, 4),
                depth.cpu().numpy().reshape(self.uImageV, self.uImageU))

    def _desc(self):
        return make_camera(self.mPosition, self.rotationDegrees, self.uImageU, self.uImageV,
                           self.uSamplesPerPixel, self.mDistToFilm, (self.sensorSizeX, self.sensorSizeY))


class MeshEngine:
    """Scene container.  `loadTriangles` takes what the adapter flattens out of
    MeshEngine::sceneMeshes in createBVH order (meshEngine.cpp:660-718) and
    builds the device BVH (the role of createBVH, meshEngine.cpp:649-724)."""

    def __init__(self, device=0):
        self.device = device
        self.sceneAccelerator = None  # vermilion_amd.Scene
        self.boundTextures = []       # float images; only [0] is sampled (pathtracer.cpp:65)

    def loadTriangles(self, pos, nrm, uv=None, spheres=None, leaf_size=4):
        if self.sceneAccelerator is not None:
            self.sceneAccelerator.close()  # meshEngine.cpp:722
        self.sceneAccelerator = Scene(pos, nrm, uv, spheres=spheres, leaf_size=leaf_size, device=self.device)
        return True

    def bindTexture(self, image):
        """MeshEngine::bindTexture (meshEngine.cpp:74-93) with the decoded float image in place of
        the file name (the OpenImageIO read stays with the host).  Returns False like the reference
        when nothing usable was given."""
        if image is None or self.sceneAccelerator is None:
            return False
        self.sceneAccelerator.bind_texture(image)
        self.boundTextures.append(np.asarray(image, np.float32))
        return True

    def RayCast(self, rayStart, rayDirection):
        """-> (hit, material, location, normal, distance, uv, colour) for one ray or a batch"""
        if self.sceneAccelerator is None:
            raise RuntimeError("RayCast before a scene was loaded")
        h = self.sceneAccelerator.raycast(rayStart, rayDirection)
        return ((h["flags"] & 1) != 0, (h["flags"] & 2) != 0, h["location"], h["normal"], h["distance"],
                h["uv"], h["colour"])


class Integrator:  # integrators.h:11-16
    def Render(self, cameraList, mEng):
        raise NotImplementedError


class PathTracer(Integrator):
    """The MI355X path tracer behind Integrator::Render (replaces
    PathTracer::Render, core/integrators/pathtracer.cpp:200-328).

    seed: the reference seeds from std::random_device (pathtracer.cpp:231); a
    seed makes frames reproducible.  early_stop / sampling default to the
    reference's behaviour."""

    def __init__(self, seed=1, early_stop=True, sampling=L.VMX_SAMPLING_PARITY, collect_counters=False):
        self.seed, self.early_stop, self.sampling = seed, early_stop, sampling
        self.collect_counters = collect_counters
        self.last_stats = []

    def Render(self, cameraList, mEng):
        self.last_stats = []
        for cam in cameraList:  # pathtracer.cpp:210
            if mEng is None or mEng.sceneAccelerator is None:
                raise RuntimeError("Render without a loaded scene")
            opts = make_opts(seed=self.seed, early_stop=self.early_stop, sampling=self.sampling,
                             collect_counters=self.collect_counters)
            frame, stats = mEng.sceneAccelerator.render(cam._desc(), opts)
            frame = frame.reshape(-1, 5)
            ch = _CHANNELS[cam.renderMode]
            if cam.renderMode == vermRenderMode.Depth:
                cam.mImage[:] = frame[:, 4]
            else:
                cam.mImage[:] = frame[:, :ch].reshape(-1)
            self.last_stats.append(stats)


class RenderEngine:
    """Orchestrator subset: plugin seam + draw (renderEngine.cpp:49-166)."""

    def __init__(self, mEng=None):
        self.mCameras = []
        self.mIntegrator = None
        self.mMeshEngine = mEng
        self.bHasMeshEng = mEng is not None
        self.log = []

    def assignIntegrator(self, integrator):  # renderEngine.cpp:70-78
        self.mIntegrator = integrator

    def assignEngine(self, mEng):  # renderEngine.cpp:109-113
        self.bHasMeshEng = True
        self.mMeshEngine = mEng

    def CreateInternalDefaultCamera(self):  # renderEngine.cpp:115-145
        if self.mCameras and self.mCameras[0] is not None:
            return
        self.mCameras.append(Camera(cameraSettings()))

    def draw(self):  # renderEngine.cpp:147-166
        if not self.bHasMeshEng:
            self.log.append("Renderer called without mesh engine")
            return
        if not self.mCameras:
            self.log.append("Renderer has no camera... Defaulting")
            self.CreateInternalDefaultCamera()
        if self.mIntegrator:
            self.mIntegrator.Render(self.mCameras, self.mMeshEngine)
