"""PlanResources (internal/ruletable/plan.go, internal/ruletable/planner): the query planner over the rule table."""
from .planner import Planner, StrictEvaluationError, plan_resources_response  # noqa: F401
