// reconstruction.hpp -- the view / camera registry PoseGraphBuilder::run fills.
//
// Minimal restatement of the types the outer API's signature names, `void run(Reconstruction&, PoseGraph&)`
// (pose_graph_builder.h:69-71), with the members that call path touches (pose_graph_builder.h:241-291):
//   reconstruction::PinholeCamera   include/pinhole_camera.h   K = [fx 0 cx; 0 fy cy; 0 0 1], width, height
//   reconstruction::View            include/view.h             camera id, view id, metadata ("name", "extension"), pose
//   reconstruction::Reconstruction  include/reconstruction.h   addCamera / addView / getCamera / getView / id lists
// Lookups of unknown ids return an "undefined" value object (the reference returns a reference to a function-local
// static; value semantics here).  Eigen::Matrix3d -> reconstruction::Matrix3d (row-major double[9]).
#pragma once
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "pose_graph_builder.hpp"

namespace reconstruction {

typedef size_t CameraId;  // include/types.h:9-14
typedef std::unordered_map<std::string, std::string> ViewMetadata;
constexpr double UndefinedCameraParameter = -1.0;
constexpr size_t UndefinedViewParameter = std::numeric_limits<size_t>::max();

class PinholeCamera {
   public:
    PinholeCamera()
        : PinholeCamera(UndefinedCameraParameter, UndefinedCameraParameter, UndefinedCameraParameter, UndefinedCameraParameter) {}
    PinholeCamera(double focal_length_x_, double focal_length_y_, double principal_point_x_, double principal_point_y_) {
        setIntrinsics(focal_length_x_, focal_length_y_, principal_point_x_, principal_point_y_);
    }
    void setIntrinsics(double focal_length_x_, double focal_length_y_, double principal_point_x_, double principal_point_y_) {
        intrinsic_parameters = Matrix3d{{focal_length_x_, 0, principal_point_x_, 0, focal_length_y_, principal_point_y_, 0, 0, 1}};
    }
    void setDimensions(double width_, double height_) {
        width = width_;
        height = height_;
    }
    void setWidth(double width_) { width = width_; }
    void setHeight(double height_) { height = height_; }
    double getWidth() const { return width; }
    double getHeight() const { return height; }
    const Matrix3d& getIntrinsics() const { return intrinsic_parameters; }

   protected:
    Matrix3d intrinsic_parameters{};
    double width = 0, height = 0;
};

class View {
   public:
    View(CameraId camera_id_ = UndefinedViewParameter, ViewId view_id_ = UndefinedViewParameter)
        : view_id(view_id_), camera_id(camera_id_) {}
    const size_t& cameraId() const { return camera_id; }
    const size_t& id() const { return view_id; }
    const Pose& getPose() const { return T_view_world; }
    Pose& getMutablePose() { return T_view_world; }
    bool hasPose() const { return has_pose; }
    void setPose(const Matrix3d& rotation_, const Vector3d& translation_) {
        T_view_world = Pose(rotation_, translation_);
        has_pose = true;
    }
    const ViewMetadata& getMetadata() const { return metadata; }
    ViewMetadata& getMutableMetadata() { return metadata; }

   protected:
    ViewId view_id;
    CameraId camera_id;
    Pose T_view_world{SE3d()};
    ViewMetadata metadata;
    bool has_pose = false;
};

class Reconstruction {
   public:
    bool addCamera(CameraId camera_id_) {
        return addCamera(camera_id_, UndefinedCameraParameter, UndefinedCameraParameter, UndefinedCameraParameter,
                         UndefinedCameraParameter);
    }
    bool addCamera(CameraId camera_id_, double focal_length_x_, double focal_length_y_, double principal_point_x_,
                   double principal_point_y_) {
        if (!cameras.emplace(camera_id_, PinholeCamera(focal_length_x_, focal_length_y_, principal_point_x_, principal_point_y_)).second)
            return false;
        camera_ids.push_back(camera_id_);
        return true;
    }
    bool addView(CameraId camera_id_, ViewId view_id_) {
        if (!views.emplace(view_id_, View(camera_id_, view_id_)).second) return false;
        view_ids.push_back(view_id_);
        return true;
    }
    bool hasCamera(CameraId id) const { return cameras.count(id) != 0; }
    bool hasView(ViewId id) const { return views.count(id) != 0; }
    PinholeCamera getCamera(CameraId camera_id_) const {
        auto it = cameras.find(camera_id_);
        return it == cameras.end() ? PinholeCamera() : it->second;
    }
    PinholeCamera& getMutableCamera(CameraId camera_id_) {
        auto it = cameras.find(camera_id_);
        return it == cameras.end() ? scratch_camera : it->second;
    }
    size_t getViewNumber() const { return views.size(); }
    const std::vector<CameraId>& getCameraIds() const { return camera_ids; }
    View getView(ViewId view_id_) const {
        auto it = views.find(view_id_);
        return it == views.end() ? View() : it->second;
    }
    View& getMutableView(ViewId view_id_) {
        auto it = views.find(view_id_);
        return it == views.end() ? scratch_view : it->second;
    }
    const std::vector<ViewId>& getViewIds() const { return view_ids; }

   protected:
    std::unordered_map<CameraId, PinholeCamera> cameras;
    std::unordered_map<ViewId, View> views;
    std::vector<CameraId> camera_ids;
    std::vector<ViewId> view_ids;
    PinholeCamera scratch_camera;  // target of writes to unknown ids (never read back)
    View scratch_view;
};

}  // namespace reconstruction
