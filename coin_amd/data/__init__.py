from .synthetic import SyntheticTwoViewLoader, SyntheticTeacherCache, synthetic_teacher_result  # noqa: F401
