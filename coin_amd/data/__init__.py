from .synthetic import SyntheticTwoViewLoader, SyntheticTeacherCache, synthetic_teacher_result  # noqa: F401
from .common import AspectRatioGroupedDatasetTwoCrop  # noqa: F401
from .dataset_mapper import DatasetMapperUnsupervised  # noqa: F401
from .build import TrainingSampler, build_detection_unsupervised_train_loader, build_unsupervised_batch_data_loader  # noqa: F401
