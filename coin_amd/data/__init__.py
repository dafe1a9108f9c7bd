from .synthetic import SyntheticTwoViewLoader, SyntheticTeacherCache, synthetic_teacher_result  # noqa: F401
from .common import AspectRatioGroupedDatasetTwoCrop  # noqa: F401
from .dataset_mapper import DatasetMapperUnsupervised, TESTMapper  # noqa: F401
from .build import (InferenceSampler, LazyTestSet, TrainingSampler, build_detection_test_loader, build_detection_unsupervised_train_loader,  # noqa: F401
                    build_unsupervised_batch_data_loader)
