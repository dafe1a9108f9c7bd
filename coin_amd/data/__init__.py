from .synthetic import SyntheticTwoViewLoader, SyntheticTeacherCache, synthetic_teacher_result  # noqa: F401
from .common import AspectRatioGroupedDatasetTwoCrop  # noqa: F401
from .dataset_mapper import DatasetMapperUnsupervised  # noqa: F401
