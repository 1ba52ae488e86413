import sys
sys.path.insert(0,"solstrale-rust_amd"); sys.path.insert(0,"tests")
from solstrale_amd import DeviceScene, RenderConfig, scenes
sc = scenes.cornell_box(RenderConfig(1920,1080,16))
ds = DeviceScene(sc); ds.kernel_timing(True); ds.render(0,16,1); print("grid", ds.last_kernel_ms())
