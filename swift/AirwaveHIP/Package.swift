// swift-tools-version:5.9
// Swift package that exposes libairwave_hip.so (C ABI, include/airwave_hip.h) to Swift host code and
// provides drop-in conformances to Airwave's own protocols.  NOT compiled in this repository's CI
// (no swiftc in the build image): shipped as the binding a maintainer adds — see INTEGRATION.md.
import PackageDescription

let package = Package(
    name: "AirwaveHIP",
    products: [.library(name: "AirwaveHIP", targets: ["AirwaveHIP"])],
    targets: [
        .systemLibrary(name: "CAirwaveHIP", path: "Sources/CAirwaveHIP"),
        .target(name: "AirwaveHIP", dependencies: ["CAirwaveHIP"], path: "Sources/AirwaveHIP"),
    ]
)
