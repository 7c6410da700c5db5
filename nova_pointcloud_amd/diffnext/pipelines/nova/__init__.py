from .pipeline_nova import NOVAPipeline  # noqa: F401
from .pipeline_utils import NOVAPipelineOutput  # noqa: F401
