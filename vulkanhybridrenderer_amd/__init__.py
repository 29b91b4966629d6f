"""MI355X-native hybrid ray-tracing hot path (RT shadows / AO / mirror reflections + SVGF) behind the
render-graph pass API of RMichelsen/VulkanHybridRenderer.  See DESIGN.md."""
__version__ = "0.1.0"
